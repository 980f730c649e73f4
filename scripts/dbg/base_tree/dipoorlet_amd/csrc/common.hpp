// Shared by every translation unit of libdipoorlet_hip.so: error state, launch constants, wave reductions,
// the order-preserving fp32 <-> u32 encoding, the streaming skeleton and the work-item -> workgroup mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/dipoorlet_hip.h"

// Bit-exact numpy parity needs every fp32 operation rounded on its own: HIP's default
// -ffp-contract=fast would fuse i*step + first into one FMA (__fmul_rn/__fadd_rn are plain * and +
// in this toolchain).  Also passed as a flag by csrc/build.py.
#pragma clang fp contract(off)

namespace dpl {
inline thread_local char g_err[512] = "";  // one per thread for the whole library (shared by every translation unit)
}

namespace {

using dpl::g_err;
constexpr int kBlock = 256;   // 4 waves of 64
#ifndef DPL_UNROLL
#define DPL_UNROLL 4
#endif
constexpr int kUnroll = DPL_UNROLL;    // float4 loads per lane per register set (two sets are in flight)
constexpr int kWave = 64;
using f4 = __attribute__((ext_vector_type(4))) float;  // native vector: nontemporal builtins need it

int fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -1;
}
int fail_msg(const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return -2;
}
#define DPL_LAUNCH_CHECK(name)                              \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return fail(name, e__);      \
    } while (0)

// ---------------------------------------------------------------- fp32 <-> order-preserving u32
__host__ __device__ inline uint32_t enc_f32(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float dec_f32(uint32_t u) {
    uint32_t b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// ---------------------------------------------------------------- the streaming skeleton
// Applies op(float) to every element of p[0..n).  Scalar head up to 16-B alignment, float4 body with
// kUnroll independent loads per lane (each wave instruction covers 1 KiB contiguous), scalar tail.
// Pointers fetched from the segment table are generic to the compiler; casting to address space 1
// makes the loads global_load_dwordx4 (flat loads would also tick the LDS counter and stall ds ops).
typedef const __attribute__((address_space(1))) f4* gptr_f4;
typedef const __attribute__((address_space(1))) float* gptr_f32;

template <class Op>
__device__ __forceinline__ void stream_span(const float* __restrict__ p_generic, uint32_t n, Op& op) {
    const uint32_t tid = threadIdx.x;
    gptr_f32 p = (gptr_f32)p_generic;
    uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)p_generic & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    if (tid < head) op(p[tid]);
    p += head;
    n -= head;
    const uint32_t nvec = n >> 2;
    gptr_f4 pv = (gptr_f4)p;
    uint32_t i = tid;
    constexpr uint32_t kStride = kUnroll * kBlock;
    // software pipeline: the next kUnroll loads are issued before the current ones are consumed, so a wave
    // always has 4-8 KiB in flight and few waves per SIMD suffice (fewer, larger work items stream
    // measurably faster from HBM than many small ones)
    // (ping-pong register sets A/B, loop unrolled by two: a register copy nxt -> cur would make the compiler
    // wait for the loads it just issued)
#define DPL_LOAD(buf, base)                                                                       \
    _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) buf[u] = __builtin_nontemporal_load(pv + (base) + u * kBlock)
#define DPL_EAT(buf)                                 \
    _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) { \
        op(buf[u].x);                                \
        op(buf[u].y);                                \
        op(buf[u].z);                                \
        op(buf[u].w);                                \
    }
    if (i + (kUnroll - 1) * kBlock < nvec) {
        f4 A[kUnroll], B[kUnroll];
        DPL_LOAD(A, i);
        i += kStride;
        for (;;) {
            if (!(i + (kUnroll - 1) * kBlock < nvec)) {
                DPL_EAT(A);
                break;
            }
            DPL_LOAD(B, i);
            i += kStride;
            DPL_EAT(A);
            if (!(i + (kUnroll - 1) * kBlock < nvec)) {
                DPL_EAT(B);
                break;
            }
            DPL_LOAD(A, i);
            i += kStride;
            DPL_EAT(B);
        }
    }
#undef DPL_LOAD
#undef DPL_EAT
    for (; i < nvec; i += kBlock) {
        f4 v = __builtin_nontemporal_load(pv + i);
        op(v.x);
        op(v.y);
        op(v.z);
        op(v.w);
    }
    const uint32_t t = (nvec << 2) + tid;
    if (t < n) op(p[t]);
}

// Work distribution shared by the streaming kernels: block b owns items [bb[b], bb[b+1]) (a balanced,
// contiguous share of the launch's elements, dpl_build_balanced_items) or, when bb is null, item b alone.
__device__ __forceinline__ void block_items(const uint32_t* __restrict__ bb, uint32_t& k0, uint32_t& k1) {
    if (bb) {
        k0 = bb[blockIdx.x];
        k1 = bb[blockIdx.x + 1];
    } else {
        k0 = blockIdx.x;
        k1 = k0 + 1;
    }
}

inline int grid_for(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

inline int check_blocks(const char* who, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks) {
    if (n_blocks <= 0 || n_blocks > 0x7FFFFFFFll || (!d_block_begin && n_blocks != n_items)) {
        snprintf(g_err, sizeof(g_err), "%s: n_blocks must be positive and equal n_items when d_block_begin is null", who);
        return -2;
    }
    return 0;
}

}  // namespace
