// A SMALL fp32 matrix product for the classifier head of a convolutional network: C = alpha * A * B + beta * bias.
//
// Why it is here at all.  The calibration forward (forward_net.py:192-237: the reference runs the ONNX graph with onnxruntime) is
// MIOpen convolutions and torch's element-wise kernels, plus — for ResNet-50 and its kind — ONE Gemm, the last node:
// [batch, 2048] x [2048, 1000].  Sending that one product to hipBLASLt makes a fresh process load the library's kernels for
// gfx950: 0.2 s on a helper thread, during which every other code object the first forward needs (MIOpen's solvers, torch's
// kernels) queues behind it — measured on MI355X (scripts/e2e_blas_ab.sh) the first forward is issued 70 - 80 ms later and a
// 1024-image calibration run (0.45 s) ends that much later.  This kernel lives in the library that is loaded anyway.
// It is NOT a GEMM library: products above 2^28 multiply-adds (a transformer's) go to hipBLASLt as before (executor.small_gemm).
//
// Shape of the work: M = batch (<= 64 by default), N = classes, K = features — 0.13 GFMA for ResNet-50, read-bound on B (8 MB).
// One workgroup of 256 threads per 32 x 64 tile of C and per SPLIT of K (a 64 x 1000 result has 32 tiles: with one workgroup per
// tile 224 of the 256 CUs idle and each tile pays 64 exposed load latencies — 200 us; K cut into 16 splits: 512 workgroups, 4
// latencies each — 24 us), K in steps of 32 through LDS; a thread owns 2 x 4 outputs.  Sums run over k in ascending order with one
// fused multiply-add per term, the splits' partial sums are added in ascending order by a second kernel, and the number of splits is
// a function of N and K only: a row of C depends on that row of A and on B — not on the launch, not on M (the batch it came in).
#include "common.hpp"
#include <algorithm>

namespace {

constexpr int kGM = 32, kGN = 64, kGK = 32;

// splits == 1: C = alpha * A B + beta * bias.  splits > 1: P[z] = A[:, kz] B[kz, :] for the z-th range of K (k_per each, a multiple of kGK).
__global__ __launch_bounds__(kBlock) void k_gemm_small(const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                        int64_t sbk, int64_t sbn, int64_t bias_sm, int64_t bias_sn, float alpha,
                                                        float beta, int k_per, int splits) {
    __shared__ float As[kGK][kGM + 1];       // [k][m]: a thread reads two neighbours along m
    __shared__ float Bs[kGK][kGN + 4];       // [k][n]: a thread reads four neighbours along n
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * kGM, n0 = blockIdx.x * kGN;
    const int tm = (tid >> 4) * 2, tn = (tid & 15) * 4;
    const int k_lo = blockIdx.z * k_per, k_hi = min(K, k_lo + k_per);
    float acc[2][4] = {};
    for (int k0 = k_lo; k0 < k_hi; k0 += kGK) {
        // A is [M, K] row-major: consecutive lanes along k
        for (int i = tid; i < kGM * kGK; i += kBlock) {
            int m = i / kGK, k = i % kGK;
            As[k][m] = (m0 + m < M && k0 + k < k_hi) ? A[(int64_t)(m0 + m) * K + k0 + k] : 0.0f;
        }
        // B(k, n) = B[k * sbk + n * sbn]: consecutive lanes along whichever index is contiguous in memory
        if (sbn == 1) {
            for (int i = tid; i < kGK * kGN; i += kBlock) {
                int k = i / kGN, n = i % kGN;
                Bs[k][n] = (k0 + k < k_hi && n0 + n < N) ? B[(int64_t)(k0 + k) * sbk + n0 + n] : 0.0f;
            }
        } else {
            for (int i = tid; i < kGK * kGN; i += kBlock) {
                int n = i / kGK, k = i % kGK;
                Bs[k][n] = (k0 + k < k_hi && n0 + n < N) ? B[(int64_t)(k0 + k) * sbk + (int64_t)(n0 + n) * sbn] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < kGK; ++k) {
            float a0 = As[k][tm], a1 = As[k][tm + 1];
            float b[4] = {Bs[k][tn], Bs[k][tn + 1], Bs[k][tn + 2], Bs[k][tn + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = __builtin_fmaf(a0, b[j], acc[0][j]);
                acc[1][j] = __builtin_fmaf(a1, b[j], acc[1][j]);
            }
        }
        __syncthreads();
    }
    float* out = splits > 1 ? C + (int64_t)blockIdx.z * M * N : C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + tm + i, n = n0 + tn + j;
            if (m < M && n < N) {
                float v = acc[i][j];
                if (splits == 1) {
                    v = alpha * v;
                    if (bias) v = v + beta * bias[m * bias_sm + n * bias_sn];
                }
                out[(int64_t)m * N + n] = v;
            }
        }
}

// C = alpha * (P[0] + P[1] + ... in this order) + beta * bias
__global__ __launch_bounds__(kBlock) void k_gemm_small_sum(const float* __restrict__ P, const float* __restrict__ bias,
                                                            float* __restrict__ C, int M, int N, int splits, int64_t bias_sm,
                                                            int64_t bias_sn, float alpha, float beta) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x, mn = (int64_t)M * N;
    if (i >= mn) return;
    float v = P[i];
    for (int z = 1; z < splits; ++z) v = v + P[z * mn + i];
    v = alpha * v;
    if (bias) v = v + beta * bias[(i / N) * bias_sm + (i % N) * bias_sn];
    C[i] = v;
}

// splits of K: enough workgroups to fill the chip when M is one row of tiles, at least 4 steps of kGK each — a function of N and K
// ONLY, so that a row of C (an image's logits) is the same sum whatever batch the image came in
int gemm_splits(int64_t n, int64_t k) {
    int64_t tiles_n = (n + kGN - 1) / kGN;
    int64_t want = tiles_n >= 256 ? 1 : (256 + tiles_n - 1) / tiles_n;
    int64_t most = k / (4 * kGK);
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, most), 64));
}

}  // namespace

extern "C" uint64_t dpl_gemm_small_workspace(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k < 0) return 0;
    int sp = gemm_splits(n, k);
    return sp > 1 ? (uint64_t)sp * (uint64_t)m * (uint64_t)n * sizeof(float) : 0;
}

extern "C" int dpl_gemm_small(const float* d_a, const float* d_b, const float* d_bias, float* d_c, int64_t m, int64_t n, int64_t k,
                              int64_t b_stride_k, int64_t b_stride_n, int64_t bias_stride_m, int64_t bias_stride_n, float alpha,
                              float beta, float* d_workspace, dpl_stream_t s) {
    if (m < 0 || n < 0 || k < 0 || m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return fail_msg("dpl_gemm_small: bad sizes");
    if (m == 0 || n == 0) return 0;
    if (!d_c || (k > 0 && (!d_a || !d_b))) return fail_msg("dpl_gemm_small: null pointer");
    // (in steps: m, n, k are each below 2^31, their product may pass 2^64 and wrap below the bound)
    const uint64_t mn = (uint64_t)m * (uint64_t)n, kk = (uint64_t)(k ? k : 1);
    if (mn > DPL_GEMM_SMALL_MAX || mn > DPL_GEMM_SMALL_MAX / kk)
        return fail_msg("dpl_gemm_small: more than DPL_GEMM_SMALL_MAX multiply-adds (this is not a GEMM library: use hipBLASLt)");
    int sp = gemm_splits(n, k);
    if (sp > 1 && !d_workspace) return fail_msg("dpl_gemm_small: this product needs dpl_gemm_small_workspace(m, n, k) bytes of workspace");
    int k_per = sp > 1 ? (int)((((k + sp - 1) / sp) + kGK - 1) / kGK * kGK) : (int)k;
    dim3 grid((unsigned)((n + kGN - 1) / kGN), (unsigned)((m + kGM - 1) / kGM), (unsigned)sp);
    if (grid.y > 65535) return fail_msg("dpl_gemm_small: too many rows");
    hipLaunchKernelGGL(k_gemm_small, grid, dim3(kBlock), 0, (hipStream_t)s, d_a, d_b, d_bias, sp > 1 ? d_workspace : d_c, (int)m, (int)n,
                       (int)k, b_stride_k, b_stride_n, bias_stride_m, bias_stride_n, alpha, beta, k_per, sp);
    DPL_LAUNCH_CHECK("dpl_gemm_small");
    if (sp > 1) {
        hipLaunchKernelGGL(k_gemm_small_sum, dim3((unsigned)((m * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)s,
                           (const float*)d_workspace, d_bias, d_c, (int)m, (int)n, sp, bias_stride_m, bias_stride_n, alpha, beta);
        DPL_LAUNCH_CHECK("dpl_gemm_small (sum)");
    }
    return 0;
}
