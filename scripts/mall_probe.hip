// Probe for the single-read OCTAV design (round 2): what does a SECOND read of recently streamed data cost on MI355X
// when it is served by the 256 MiB Infinity Cache (MALL) or the XCD L2s instead of HBM?
//   E1  re-read bandwidth of a buffer of S MB (plain / nt loads)
//   E2  windowed two-pass over 3.4 GB: [read w ; read w] per window (plain-plain, plain-nt, nt-nt)
//   E3  software-pipelined two-pass: read(w+1) ; reread(w)   (working set 2 windows)
//   E4  one launch, half the workgroups stream from HBM, half re-read a MALL-sized buffer (do the two add up?)
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mall_probe scripts/mall_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) f4* gptr_f4;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ p, uint64_t nvec, float* __restrict__ out) {
    gptr_f4 pv = (gptr_f4)p;
    // contiguous share per block, like the product's balanced partition
    uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
    per = (per + 1023) & ~1023ull;
    uint64_t b0 = per * blockIdx.x, b1 = b0 + per;
    if (b1 > nvec) b1 = nvec;
    float acc = 0.f;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 1024) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint64_t j = i + u * 256;
            if (j < b1) v[u] = NT ? __builtin_nontemporal_load(pv + j) : pv[j];
            else v[u] = f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    if (acc == 12345.678f) out[0] = acc;
}

// E4: blocks with (blockIdx.x & 1) stream `hbm` once; the others re-read `hot` `reps` times
template <bool NT_HBM>
__global__ __launch_bounds__(256) void k_mix(const float* __restrict__ hbm, uint64_t nvec_hbm, const float* __restrict__ hot,
                                            uint64_t nvec_hot, int reps, float* __restrict__ out) {
    const uint32_t half = gridDim.x >> 1, me = blockIdx.x >> 1;
    float acc = 0.f;
    if (blockIdx.x & 1) {
        gptr_f4 pv = (gptr_f4)hbm;
        uint64_t per = ((nvec_hbm + half - 1) / half + 1023) & ~1023ull;
        uint64_t b0 = per * me, b1 = b0 + per;
        if (b1 > nvec_hbm) b1 = nvec_hbm;
        for (uint64_t i = b0 + threadIdx.x; i < b1; i += 256) {
            f4 v = NT_HBM ? __builtin_nontemporal_load(pv + i) : pv[i];
            acc = fmaxf(acc, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        gptr_f4 pv = (gptr_f4)hot;
        uint64_t per = ((nvec_hot + half - 1) / half + 1023) & ~1023ull;
        uint64_t b0 = per * me, b1 = b0 + per;
        if (b1 > nvec_hot) b1 = nvec_hot;
        for (int r = 0; r < reps; ++r)
            for (uint64_t i = b0 + threadIdx.x; i < b1; i += 256) {
                f4 v = pv[i];
                acc = fmaxf(acc, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
    }
    if (acc == 12345.678f) out[0] = acc;
}

static float* d_out;
static hipStream_t st;

static void launch_read(const float* p, uint64_t bytes, bool nt, int grid) {
    uint64_t nvec = bytes / 16;
    if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, st, p, nvec, d_out);
    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, st, p, nvec, d_out);
}

int main() {
    CK(hipStreamCreate(&st));
    const uint64_t MB = 1ull << 20;
    const uint64_t total = 3400 * MB;
    float* buf;
    CK(hipMalloc(&buf, total + 512 * MB));
    CK(hipMalloc(&d_out, 64));
    // random-ish fill (non-zero, DVFS-realistic)
    {
        std::vector<float> h(64 * MB / 4);
        uint32_t s = 12345;
        for (auto& x : h) {
            s = s * 1664525u + 1013904223u;
            x = (float)(int32_t)s * 4.6e-10f;
        }
        for (uint64_t o = 0; o < total + 512 * MB; o += 64 * MB) {
            uint64_t len = total + 512 * MB - o < 64 * MB ? total + 512 * MB - o : 64 * MB;
            CK(hipMemcpy((char*)buf + o, h.data(), len, hipMemcpyHostToDevice));
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, int reps) {
        fn();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };

    printf("== E1: re-read of an S MB buffer (grid 1024)\n");
    for (int nt = 0; nt < 2; ++nt)
        for (uint64_t S : {8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 3400}) {
            float ms = timeit([&] { launch_read(buf, S * MB, nt, 1024); }, S <= 256 ? 40 : 8);
            printf("E1 nt=%d S=%4llu MB  %.4f ms  %.2f TB/s\n", nt, (unsigned long long)S, ms, S * MB / ms * 1e-9);
        }
    printf("== E1b: grid size sweep on S=64 MB and S=3400 MB (plain)\n");
    for (int grid : {256, 512, 1024, 2048, 4096})
        for (uint64_t S : {64, 3400}) {
            float ms = timeit([&] { launch_read(buf, S * MB, false, grid); }, S <= 256 ? 40 : 8);
            printf("E1b grid=%d S=%4llu MB  %.4f ms  %.2f TB/s\n", grid, (unsigned long long)S, ms, S * MB / ms * 1e-9);
        }

    printf("== E2: windowed two-pass over 3400 MB: per window [read ; reread]; credited = 3400 MB once\n");
    for (int mode = 0; mode < 3; ++mode)  // 0: plain,plain  1: plain,nt  2: nt,nt
        for (uint64_t W : {16, 32, 64, 100, 128, 200, 425, 850, 3400}) {
            float ms = timeit(
                [&] {
                    for (uint64_t o = 0; o < total; o += W * MB) {
                        uint64_t len = (o + W * MB <= total) ? W * MB : total - o;
                        launch_read((const float*)((char*)buf + o), len, mode == 2, 1024);
                        launch_read((const float*)((char*)buf + o), len, mode >= 1, 1024);
                    }
                },
                4);
            printf("E2 mode=%d W=%4llu MB  %.3f ms  credited %.2f TB/s (two HBM reads would be %.2f)\n", mode,
                   (unsigned long long)W, ms, total / ms * 1e-9, 0.5 * 6.3);
        }
    printf("== E3: pipelined: read(w+1) ; reread(w)\n");
    for (int mode = 0; mode < 2; ++mode)
        for (uint64_t W : {16, 32, 64, 100, 128}) {
            float ms = timeit(
                [&] {
                    uint64_t prev_o = 0, prev_len = 0;
                    for (uint64_t o = 0; o < total; o += W * MB) {
                        uint64_t len = (o + W * MB <= total) ? W * MB : total - o;
                        launch_read((const float*)((char*)buf + o), len, false, 1024);
                        if (prev_len) launch_read((const float*)((char*)buf + prev_o), prev_len, mode == 1, 1024);
                        prev_o = o;
                        prev_len = len;
                    }
                    launch_read((const float*)((char*)buf + prev_o), prev_len, mode == 1, 1024);
                },
                4);
            printf("E3 mode=%d W=%4llu MB  %.3f ms  credited %.2f TB/s\n", mode, (unsigned long long)W, ms, total / ms * 1e-9);
        }
    printf("== E4: one launch: half the blocks stream 3400 MB from HBM, half re-read a hot buffer\n");
    for (int nt = 0; nt < 2; ++nt)
        for (uint64_t H : {16, 64, 128}) {
            for (int reps : {0, 10, 25, 50}) {
                const float* hot = (const float*)((char*)buf + total);
                float ms = timeit(
                    [&] {
                        if (nt) hipLaunchKernelGGL(k_mix<true>, dim3(2048), dim3(256), 0, st, buf, total / 16, hot, H * MB / 16, reps, d_out);
                        else hipLaunchKernelGGL(k_mix<false>, dim3(2048), dim3(256), 0, st, buf, total / 16, hot, H * MB / 16, reps, d_out);
                    },
                    4);
                printf("E4 nt_hbm=%d hot=%3llu MB reps=%2d  %.3f ms  hbm-stream %.2f TB/s  hot %.2f TB/s  sum %.2f TB/s\n", nt,
                       (unsigned long long)H, reps, ms, total / ms * 1e-9, (double)H * MB * reps / ms * 1e-9,
                       (total + (double)H * MB * reps) / ms * 1e-9);
            }
        }
    return 0;
}
