// How fast can a 1/16 SAMPLE of a 3.4 GB buffer be read on MI355X, by the shape of the sample?  (k_octav_probe reads one
// 128-byte line of every 2 KiB: 213 MB in ~70 us = 3 TB/s of sampled bytes.)  Patterns: L lines per window of L x 2 KiB,
// adjacent (one run of L x 128 bytes at a hashed offset) or spread (L hashed offsets inside the window, or inside its first
// `sub` bytes).  Eight lanes read one line (16 bytes each), 8 lines in flight per lane group, like the product kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/sample_probe scripts/sample_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

using f4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) f4* gptr_f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: adjacent, 1: spread over the window, 2: spread over the first `sub` bytes of the window
__global__ __launch_bounds__(256, 6) void k_sample(const float* __restrict__ p, uint64_t n_lines, uint32_t L, uint32_t win_bytes, int mode,
                                                   uint32_t sub, uint32_t salt, float* __restrict__ out) {
    const char* base = (const char*)p;
    const uint32_t sublane = threadIdx.x & 7u;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 3, n_groups = ((uint64_t)gridDim.x * 256) >> 3;
    float acc = 0.f;
    for (uint64_t g0 = group; g0 < n_lines; g0 += n_groups * 8) {
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t g = g0 + (uint64_t)k * n_groups;
            if (g < n_lines) {
                const uint64_t w = g / L;
                const uint32_t j = (uint32_t)(g % L);
                uint32_t off;
                if (mode == 0) {
                    const uint32_t slots = win_bytes / 128 - (L - 1);
                    off = (mix((uint32_t)w ^ salt) % slots + j) * 128;
                } else {
                    const uint32_t span = mode == 2 ? sub : win_bytes;
                    const uint32_t per = span / 128 / L;   // a sub-range of slots per line: distinct lines
                    off = (j * per + mix(((uint32_t)w * 7u + j) ^ salt) % per) * 128;
                }
                v[k] = __builtin_nontemporal_load((gptr_f4)(base + w * win_bytes + off) + sublane);
            } else {
                v[k] = f4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaxf(acc, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const uint64_t bytes = 3404592128ull;
    float *d, *d_out;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&d_out, 4));
    CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Pat { const char* name; uint32_t L, win; int mode; uint32_t sub; };
    const Pat pats[] = {
        {"1 line / 2 KiB (product)", 1, 2048, 0, 0},
        {"2 adjacent / 4 KiB", 2, 4096, 0, 0},
        {"2 spread / 4 KiB", 2, 4096, 1, 0},
        {"2 within first 1 KiB / 4 KiB", 2, 4096, 2, 1024},
        {"2 within first 2 KiB / 4 KiB", 2, 4096, 2, 2048},
        {"4 adjacent / 8 KiB", 4, 8192, 0, 0},
        {"4 spread / 8 KiB", 4, 8192, 1, 0},
        {"4 within first 2 KiB / 8 KiB", 4, 8192, 2, 2048},
        {"4 within first 4 KiB / 8 KiB", 4, 8192, 2, 4096},
        {"8 adjacent / 16 KiB", 8, 16384, 0, 0},
        {"8 within first 4 KiB / 16 KiB", 8, 16384, 2, 4096},
        {"1 line / 1 KiB (rate 1/8)", 1, 1024, 0, 0},
        {"1 line / 4 KiB (rate 1/32)", 1, 4096, 0, 0},
    };
    for (const Pat& q : pats) {
        const uint64_t n_lines = bytes / q.win * q.L;
        for (int grid : {1536, 3072}) {
            for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_sample, dim3(grid), dim3(256), 0, 0, d, n_lines, q.L, q.win, q.mode, q.sub, 0x9e3779b9u * (100 + r), d_out);
            CK(hipEventRecord(e0));
            const int reps = 10;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_sample, dim3(grid), dim3(256), 0, 0, d, n_lines, q.L, q.win, q.mode, q.sub, 0x9e3779b9u * (r + 1 + grid), d_out);   // (other lines every time: 213 MB would sit in the Infinity Cache)
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            printf("%-34s grid %4d: %7.1f us  %6.0f GB/s of sampled bytes (%.0f MB)\n", q.name, grid, ms * 1e3, n_lines * 128.0 / ms / 1e6, n_lines * 128.0 / 1e6);
        }
    }
    return 0;
}
