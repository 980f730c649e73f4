// MI355X (gfx950 / CDNA4) activation-calibration kernels + C ABI (include/dipoorlet_hip.h).
//
// Everything here is an HBM-bound streaming reduction / scatter-add: no MFMA.  Design rules
// (guides: cdna_hip_programming.md G2/G11/G12/G13, MI355X_MICROARCH.md §LDS/§HBM):
//   * 16 B per lane coalesced loads (global_load_dwordx4), several independent loads in flight,
//     one workgroup per work item (a contiguous chunk of ONE tensor), >> 256 workgroups per launch;
//   * wave64 reductions with DPP/ds_bpermute shuffles, then a tiny LDS combine per workgroup;
//   * histograms privatised in LDS (ds_add_u32), exact zeros counted in registers (ReLU outputs are
//     ~50 % zeros and would otherwise serialise on one LDS address), one flush per workgroup;
//   * order-encoded integer atomics for fp32 min/max, so accumulators persist across launches.
#include "common.hpp"

// Bit-exact numpy parity needs every fp32 operation rounded on its own: HIP's default
// -ffp-contract=fast would fuse i*step + first into one FMA (__fmul_rn/__fadd_rn are plain * and +
// in this toolchain).  Also passed as a flag by csrc/build.py.
#pragma clang fp contract(off)

namespace {

// ================================================================ K1: running min / max
struct MinMaxOp {
    float mn, mx;
    uint32_t nan;
    __device__ __forceinline__ void operator()(float x) {
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
        nan |= (x != x);
    }
};

__global__ __launch_bounds__(kBlock) void k_minmax(const dpl_work_item* __restrict__ items,
                                                    const uint32_t* __restrict__ bb,
                                                    const float* const* __restrict__ segs,
                                                    uint32_t* __restrict__ min_enc, uint32_t* __restrict__ max_enc,
                                                    uint32_t* __restrict__ nan_flag) {
    __shared__ float s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    __shared__ uint32_t s_nan[kBlock / kWave];
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        MinMaxOp op{INFINITY, -INFINITY, 0u};
        stream_span(segs[it.seg] + it.offset, it.count, op);
        float mn = wave_min(op.mn), mx = wave_max(op.mx);
        uint32_t nn = __any(op.nan) ? 1u : 0u;
        const int w = threadIdx.x / kWave;
        if ((threadIdx.x & (kWave - 1)) == 0) {
            s_mn[w] = mn;
            s_mx[w] = mx;
            s_nan[w] = nn;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 1; j < kBlock / kWave; ++j) {
                mn = fminf(mn, s_mn[j]);
                mx = fmaxf(mx, s_mx[j]);
                nn |= s_nan[j];
            }
            if (mn <= mx) {  // false only when the chunk held nothing but NaN
                atomicMin(min_enc + it.slot, enc_f32(mn));
                atomicMax(max_enc + it.slot, enc_f32(mx));
            }
            if (nn) atomicOr(nan_flag + it.slot, 1u);
        }
        __syncthreads();
    }
}

__global__ void k_minmax_init(uint32_t* mn, uint32_t* mx, uint32_t* nan, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mn[i] = 0xFFFFFFFFu;
        mx[i] = 0u;
        nan[i] = 0u;
    }
}

__global__ void k_minmax_finalize(const uint32_t* mn, const uint32_t* mx, const uint32_t* nan, int64_t n,
                                  float* omn, float* omx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bool bad = nan[i] != 0u || mn[i] == 0xFFFFFFFFu;
        omn[i] = bad ? NAN : dec_f32(mn[i]);
        omx[i] = bad ? NAN : dec_f32(mx[i]);
    }
}

__global__ void k_minmax_encode(const float* mn, const float* mx, int64_t n, uint32_t* emn, uint32_t* emx,
                                uint32_t* nan) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bool bad = (mn[i] != mn[i]) || (mx[i] != mx[i]);
        emn[i] = bad ? 0xFFFFFFFFu : enc_f32(mn[i]);
        emx[i] = bad ? 0u : enc_f32(mx[i]);
        nan[i] = bad ? 1u : 0u;
    }
}

// ================================================================ K2: |x| histogram, numpy-exact
// numpy's uniform-bin fast path lands every kept value a in the unique bin i with
// edge[i] <= a < edge[i+1] (last bin closed), edge[i] = fl32(fl32(i*step) + first): an index
// estimate followed by one decrement test and one increment test against those edges.  Any estimate
// within +-1 of the true bin gives the same answer, so the estimate here is a multiply by the
// reciprocal (error << 1 bin) unless the range is so small that the reciprocal is not finite.
__device__ __forceinline__ float hist_edge(int i, float step, float first) {
    return __fadd_rn(__fmul_rn((float)i, step), first);  // no FMA contraction: numpy rounds twice
}

// Two bin paths:
//   kFast  (first == 0, reciprocal finite — every non-degenerate range): `inv` carries a +1e-6 relative
//          bias (k_hist_prepare), which dominates the ~3e-7 of accumulated fp32 rounding in the estimate
//          and in the edges, so floor(a*inv) is the true bin or the one above it, never below: ONE
//          decrement test against edge(i) = fl32(i*step) settles it (the bias is < 0.02 bin at 16384 bins).
//   exact  (degenerate (-0.5, 0.5) range of an all-zero tensor, or a range so small that the reciprocal
//          overflows): numpy's own sequence — correctly rounded divide, decrement test, increment test.
// (Measured alternatives that lost and were removed: unconditional ds_add into per-lane dummy slots +2 %;
// ablations: no flush -2.5 %, no LDS atomics -3 % — the kernel is within 5 % of the plain streaming read.)
template <bool kFast>
struct HistOp {
    uint32_t* lds;
    float first, last, step, inv, denom;
    int last_bin;  // bins - 1
    float fbins;
    uint32_t nonzero;  // count of a != 0 (NaN included); exact zeros = elements - nonzero
    __device__ __forceinline__ void operator()(float x) {
        const float a = fabsf(x);
        const bool nz = (a != 0.0f);
        nonzero += nz;
#ifndef DPL_HIST_PLAIN
        if (kFast) {
            // (11 vector instructions per element where the form below takes 13 — what matters once the chip runs warm and
            // the shader clock comes down: DESIGN 3e.  No clamp: a <= last gives a * inv <= bins * (1 + 1e-6) + rounding, the
            // estimate is at most `bins`, and counter `bins` — a == last where bins * step rounds to last or below — is folded
            // into the closed last bin at the flush; the byte address in one shift-add.)
            if (nz && (a <= last)) {
                const int i = (int)__fmul_rn(a, inv);      // (v_cvt_i32_f32 maps NaN to 0: dropped by the test above)
                const uint32_t dec = (a < __fmul_rn((float)i, step)) ? 0xFFFFFFFCu : 0u;
                atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds) + (((uint32_t)i << 2) + dec)), 1u);
            }
            return;
        }
#endif
        int i;
        if (kFast) {
            i = (int)__fmul_rn(a, inv);  // v_cvt_i32_f32 saturates and maps NaN to 0
            i = i > last_bin ? last_bin : i;
            i -= (a < __fmul_rn((float)i, step)) ? 1 : 0;
        } else {
            i = (int)__fmul_rn(__fdiv_rn(__fsub_rn(a, first), denom), fbins);
            i = i > last_bin ? last_bin : i;
            i = i < 0 ? 0 : i;
            i -= (a < hist_edge(i, step, first)) ? 1 : 0;
            i += (i != last_bin && a >= hist_edge(i + 1, step, first)) ? 1 : 0;
        }
        // exact zeros are counted in a register and added to their bin once per wave (ReLU outputs are ~50 %
        // zeros: they would serialise on one LDS address); out-of-range values and NaN (a <= last false) drop;
        // a >= first always holds since first <= 0 <= a.
        if (nz && (a <= last)) atomicAdd(lds + i, 1u);  // ds_add_u32 (no return)
    }
};

template <bool kFast>
__device__ __forceinline__ void hist_body(const dpl_work_item& it, const float* const* __restrict__ segs,
                                          const dpl_hist_range& r, int bins, uint64_t* __restrict__ hist,
                                          uint32_t* lds, uint32_t* s_nz) {
    HistOp<kFast> op;
    op.lds = lds;
    op.first = r.first;
    op.last = r.last;
    op.step = r.step;
    op.inv = r.inv;
    op.denom = __fsub_rn(r.last, r.first);
    op.last_bin = bins - 1;
    op.fbins = (float)bins;
    op.nonzero = 0u;
    stream_span(segs[it.seg] + it.offset, it.count, op);
    const uint32_t nzw = wave_sum(op.nonzero);
    if ((threadIdx.x & (kWave - 1)) == 0) s_nz[threadIdx.x / kWave] = nzw;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t nzb = 0;
        for (int k = 0; k < kBlock / kWave; ++k) nzb += s_nz[k];
        // |0| is kept iff first <= 0 <= last, which always holds for a finite range
        const uint32_t z = it.count - nzb;
        if (z) atomicAdd(lds + r.zero_bin, z);
        const uint32_t top = lds[bins];        // (estimates of `bins`: values equal to `last`)
        if (top) atomicAdd(lds + bins - 1, top);
    }
    __syncthreads();
    uint64_t* __restrict__ out = hist + (uint64_t)it.slot * (uint64_t)bins;
    for (int b = threadIdx.x; b < bins; b += kBlock) {
        const uint32_t c = lds[b];
        if (c) atomicAdd(reinterpret_cast<unsigned long long*>(out + b), (unsigned long long)c);
    }
}

__global__ __launch_bounds__(kBlock) void k_abs_hist(const dpl_work_item* __restrict__ items,
                                                      const uint32_t* __restrict__ bb,
                                                      const float* const* __restrict__ segs,
                                                      const dpl_hist_range* __restrict__ ranges, int bins,
                                                      uint64_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // bins + 1 counters + one word per wave
    uint32_t* s_nz = lds + bins + 1;
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        const dpl_hist_range r = ranges[it.slot];
        if (r.status != 0u) continue;  // reference raises for this tensor; host reports it (uniform branch)
        for (int b = threadIdx.x; b <= bins; b += kBlock) lds[b] = 0u;
        __syncthreads();
        if (r.exact_div)
            hist_body<false>(it, segs, r, bins, hist, lds, s_nz);
        else
            hist_body<true>(it, segs, r, bins, hist, lds, s_nz);
        __syncthreads();
    }
}

__device__ __forceinline__ float py_max(float a, float b) { return (b > a) ? b : a; }  // python max(a, b)
__device__ __forceinline__ float py_min(float a, float b) { return (b < a) ? b : a; }  // python min(a, b)

__global__ void k_hist_prepare(const float* __restrict__ gmin, const float* __restrict__ gmax, int64_t n, int bins,
                               dpl_hist_range* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dpl_hist_range r;
    // forward_net.py:266 — data_max = max(np.max(maxlist), -np.min(minlist))
    const float dmax = py_max(gmax[i], -gmin[i]);
    float first = 0.0f, last = dmax;
    r.dmax = dmax;
    r.status = 0u;
    if (!(fabsf(last) <= 3.402823466e+38f) || last < first) r.status = 1u;  // NaN, inf (or negative) range
    if (first == last) {  // numpy _get_outer_edges: expand an empty range
        first = -0.5f;
        last = 0.5f;
    }
    const float delta = __fsub_rn(last, first);
    const float fb = (float)bins;
    r.first = first;
    r.last = last;
    r.step = __fdiv_rn(delta, fb);
    // +1e-6 relative bias: see HistOp (kFast).  1.000001f = 1 + 8*2^-23 exactly representable enough:
    // the product is rounded once more, still >= (1 + 9e-7) * bins/delta.
    r.inv = __fmul_rn(__fdiv_rn(fb, delta), 1.000001f);
    // linspace must give strictly increasing fp32 edges, else numpy raises "Too many bins"
    if (r.status == 0u) {
        const float e1 = hist_edge(1, r.step, first);
        const float el = hist_edge(bins - 1, r.step, first);
        const float el2 = hist_edge(bins - 2 > 0 ? bins - 2 : 0, r.step, first);
        if (!(r.step > 0.0f) || !(e1 > first) || !(last > el) || (bins > 2 && !(el > el2))) r.status = 2u;
    }
    r.exact_div = (first != 0.0f || !(fabsf(r.inv) <= 3.402823466e+38f) || r.step < 1.0e-30f) ? 1u : 0u;
    // bin of |x| == 0
    {
        const float a = 0.0f;
        float t = __fmul_rn(__fdiv_rn(__fsub_rn(a, first), delta), fb);
        int b = (int)t;
        b = b > bins - 1 ? bins - 1 : b;
        b = b < 0 ? 0 : b;
        if (a < hist_edge(b, r.step, first)) --b;
        if (b != bins - 1 && a >= hist_edge(b + 1, r.step, first)) ++b;
        r.zero_bin = (uint32_t)(b < 0 ? 0 : b);
    }
    out[i] = r;
}

// ================================================================ K4: percentile clip (basic_algorithm.py:40-53)
// One wave per slot.  The cumulative sum is a SEQUENTIAL fp64 accumulation in bin order (the >=
// threshold test is order sensitive), so lanes load 64 bins at a time and the wave walks them in order
// (2048 dependent fp64 additions: ~10 us per launch; through ds_bpermute shuffles and a branch per bin it was 190 us).
__global__ __launch_bounds__(kWave) void k_hist_percentile(const uint64_t* __restrict__ hist,
                                                            const float* __restrict__ gmin_a,
                                                            const float* __restrict__ gmax_a, int bins,
                                                            double threshold, float* __restrict__ clip) {
    const int slot = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t* h = hist + (uint64_t)slot * bins;
    unsigned long long tot = 0;
    for (int b = lane; b < bins; b += kWave) tot += h[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, kWave);
    const double total = (double)(long long)tot;  // int64 -> float64
    const float gmin = gmin_a[slot], gmax = gmax_a[slot];
    double accum = 0.0;
    int found = -1;
    // One chunk of 64 bins per round.  The next chunk's counts are requested BEFORE this chunk is walked (the load's latency
    // sits beside the 64 additions); the lane index of the walk is wave-uniform (v_readlane, not a shuffle through LDS) and the
    // chain of additions carries no branch: the bins that reach the threshold are collected in a mask, its lowest bit is the answer.
    uint64_t raw_next = lane < bins ? h[lane] : 0ull;
    for (int base = 0; base < bins && found < 0; base += kWave) {
        const int b = base + lane;
        // hist.astype(float32) / hist.sum()  -> float64(float32(count)) / float64(total)
        const double hv = (b < bins) ? (double)(float)(long long)raw_next / total : 0.0;
        raw_next = (b + kWave < bins) ? h[b + kWave] : 0ull;
        const int lim = (bins - base) < kWave ? (bins - base) : kWave;
        const int h_lo = __double2loint(hv), h_hi = __double2hiint(hv);
        uint64_t reached = 0;
        if (lim == kWave) {
#pragma unroll
            for (int j = 0; j < kWave; ++j) {
                accum += __hiloint2double(__builtin_amdgcn_readlane(h_hi, j), __builtin_amdgcn_readlane(h_lo, j));
                reached |= (accum >= threshold) ? (1ull << j) : 0ull;
            }
        } else {
            for (int j = 0; j < lim; ++j) {
                accum += __hiloint2double(__builtin_amdgcn_readlane(h_hi, j), __builtin_amdgcn_readlane(h_lo, j));
                reached |= (accum >= threshold) ? (1ull << j) : 0ull;
            }
        }
        if (reached) found = base + __builtin_ctzll(reached);
    }
    if (lane == 0) {
        float lo = gmin, hi = gmax;
        if (found >= 0) {
            const float dmax = py_max(-gmin, gmax);  // basic_algorithm.py:42
            const float cv = __fmul_rn((float)found + 0.5f, __fdiv_rn(dmax, (float)bins));
            lo = py_max(-cv, gmin);
            hi = py_min(cv, gmax);
        }
        clip[2 * slot] = lo;
        clip[2 * slot + 1] = hi;
    }
}

// ================================================================ K5: per-row min / max of a [rows, cols] matrix
__global__ __launch_bounds__(kBlock) void k_rowwise_minmax(const float* __restrict__ w, int64_t cols,
                                                            float* __restrict__ omn, float* __restrict__ omx) {
    __shared__ float s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    __shared__ uint32_t s_nan[kBlock / kWave];
    const float* p = w + (int64_t)blockIdx.x * cols;
    MinMaxOp op{INFINITY, -INFINITY, 0u};
    // rows can be longer than 2^32 only in theory; weights are at most a few 10^7 elements
    stream_span(p, (uint32_t)cols, op);
    float mn = wave_min(op.mn), mx = wave_max(op.mx);
    uint32_t nn = __any(op.nan) ? 1u : 0u;
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_mn[wv] = mn;
        s_mx[wv] = mx;
        s_nan[wv] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / kWave; ++k) {
            mn = fminf(mn, s_mn[k]);
            mx = fmaxf(mx, s_mx[k]);
            nn |= s_nan[k];
        }
        omn[blockIdx.x] = nn ? NAN : mn;
        omx[blockIdx.x] = nn ? NAN : mx;
    }
}

// ================================================================ K6: fused quantize -> dequantize
__device__ __forceinline__ float fq_one(float x, float scale, float zp, float qlo, float qhi) {
    float q = __fadd_rn(rintf(__fdiv_rn(x, scale)), zp);  // round half to even, then zero point
    q = fminf(fmaxf(q, qlo), qhi);                        // saturate
    return __fmul_rn(__fsub_rn(q, zp), scale);
}

// What the producer of a fake-quantised tensor would have written, applied on the way in (the reference's merge-ReLU rule puts
// most activation Q/DQ pairs directly behind a ReLU, quantize.py:50-55): kFqPreNone x; kFqPreRelu torch.relu(x) = np.maximum(x, 0)
// (NaN stays NaN; -0 and +0 quantise alike); kFqPreAddRelu relu(x + x2), the residual Add of a bottleneck and its ReLU (one fp32
// addition, rounded to nearest, as torch.add).
enum { kFqPreNone = 0, kFqPreRelu = 1, kFqPreAddRelu = 2 };
template <int PRE>
__device__ __forceinline__ float fq_pre(float x, float x2) {
    if (PRE == kFqPreAddRelu) x = __fadd_rn(x, x2);
    if (PRE != kFqPreNone) x = x < 0.f ? 0.f : x;
    return x;
}

// One workgroup fake-quantises elements [e0, e0 + cnt) of a tensor viewed as [outer, n_channels, inner] (n_channels == 1: per
// tensor).  A CONTIGUOUS chunk per workgroup (few large equal shares stream faster from HBM than a grid-stride walk), four
// 16-byte vectors per lane in flight, non-temporal loads and stores (each byte is touched once).  The channel of a vector needs no
// division in the loop: a lane's (column, channel) advance by a constant per step — 1024 elements = (1024 / inner) rows and
// (1024 % inner) columns, both computed once per chunk on the scalar unit — with one conditional wrap each.
template <int PRE>
__device__ __forceinline__ void fq_span(const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ y, uint64_t e0,
                                        uint32_t cnt, const float* __restrict__ scale_p, const int32_t* __restrict__ zp_p,
                                        uint32_t n_channels, uint32_t inner, float qlo, float qhi) {
    typedef __attribute__((address_space(1))) f4* gptr_f4w;
    const uint32_t tid = threadIdx.x;
    const float* xs = x + e0;
    const float* x2s = PRE == kFqPreAddRelu ? x2 + e0 : xs;
    float* ys = y + e0;
    // 16-byte vectors whatever the rows' length: a vector of a row that is no multiple of four long (7 x 7 maps: 49) may straddle two
    // channels — it carries the parameters of both and picks per element (rows shorter than a vector: element by element)
    const bool vec = ((((uintptr_t)xs | (uintptr_t)x2s | (uintptr_t)ys) & 15u) == 0u) && (n_channels == 1u || inner >= 4u);
    if (!vec) {   // unaligned views / rows shorter than a vector: element by element, same bookkeeping
        const uint64_t e = e0 + tid;
        uint32_t col = (uint32_t)(e % inner), c = (uint32_t)((e / inner) % n_channels);
        const uint32_t step_cols = (uint32_t)kBlock % inner, step_ch = ((uint32_t)kBlock / inner) % n_channels;
        for (uint32_t i = tid; i < cnt; i += kBlock) {
            ys[i] = fq_one(fq_pre<PRE>(xs[i], x2s[i]), scale_p[c], (float)zp_p[c], qlo, qhi);
            col += step_cols;
            c += step_ch;
            if (col >= inner) {
                col -= inner;
                c += 1u;
            }
            if (c >= n_channels) c -= n_channels;
        }
        return;
    }
    const uint32_t nvec = cnt >> 2;
    gptr_f4 xv = (gptr_f4)xs;
    gptr_f4 x2v = (gptr_f4)x2s;
    gptr_f4w yv = (gptr_f4w)ys;
    // Two register sets in rotation (as stream_span): the NEXT four vectors of a lane — and, per channel, their parameters — are
    // requested before the current four are computed and stored: eight loads in flight per lane, and a parameter look-up never
    // sits between a vector's arrival and its use.
    const bool per_channel = n_channels != 1u;
    uint32_t col = 0u, c = 0u, step_cols = 0u, step_ch = 0u;
    if (per_channel) {   // the lane's first vector: one division; then (col, c) advance by the per-step constants
        const uint64_t e = e0 + 4ull * tid;
        col = (uint32_t)(e % inner);
        c = (uint32_t)((e / inner) % n_channels);
        step_cols = (4u * kBlock) % inner;
        step_ch = ((4u * kBlock) / inner) % n_channels;
    }
    const float sc1 = scale_p[0], zp1 = (float)zp_p[0];
    const bool straddle = per_channel && (inner & 3u) != 0u;   // (uniform) a vector may end in the next channel's row
    struct Set {
        f4 v[4];
        f4 w[PRE == kFqPreAddRelu ? 4 : 1];   // the second operand of the residual Add
        float sc[4], sc2[4];
        int32_t zp[4], zp2[4];
        uint32_t left[4];   // elements of the vector that still belong to the first channel's row (>= 4: all of them)
    };
    auto load = [&](Set& st, uint32_t i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            st.v[u] = i0 + u * kBlock < nvec ? __builtin_nontemporal_load(xv + i0 + u * kBlock) : f4{0.f, 0.f, 0.f, 0.f};
            if (PRE == kFqPreAddRelu)
                st.w[PRE == kFqPreAddRelu ? u : 0] =
                    i0 + u * kBlock < nvec ? __builtin_nontemporal_load(x2v + i0 + u * kBlock) : f4{0.f, 0.f, 0.f, 0.f};
            if (per_channel) {   // (uniform)
                st.sc[u] = scale_p[c];
                st.zp[u] = zp_p[c];
                if (straddle) {
                    const uint32_t cn = c + 1u < n_channels ? c + 1u : 0u;
                    st.sc2[u] = scale_p[cn];
                    st.zp2[u] = zp_p[cn];
                    st.left[u] = inner - col;
                }
                col += step_cols;
                c += step_ch;
                if (col >= inner) {
                    col -= inner;
                    c += 1u;
                }
                if (c >= n_channels) c -= n_channels;
            }
        }
    };
    auto eat = [&](Set& st, uint32_t i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kBlock < nvec) {
                const float sc = per_channel ? st.sc[u] : sc1, zp = per_channel ? (float)st.zp[u] : zp1;
                if (PRE != kFqPreNone) {
                    const f4 w = st.w[PRE == kFqPreAddRelu ? u : 0];
                    st.v[u].x = fq_pre<PRE>(st.v[u].x, w.x);
                    st.v[u].y = fq_pre<PRE>(st.v[u].y, w.y);
                    st.v[u].z = fq_pre<PRE>(st.v[u].z, w.z);
                    st.v[u].w = fq_pre<PRE>(st.v[u].w, w.w);
                }
                if (straddle) {   // (uniform)
                    const float scb = st.sc2[u], zpb = (float)st.zp2[u];
                    const uint32_t l = st.left[u];
                    st.v[u].x = fq_one(st.v[u].x, sc, zp, qlo, qhi);
                    st.v[u].y = fq_one(st.v[u].y, l > 1u ? sc : scb, l > 1u ? zp : zpb, qlo, qhi);
                    st.v[u].z = fq_one(st.v[u].z, l > 2u ? sc : scb, l > 2u ? zp : zpb, qlo, qhi);
                    st.v[u].w = fq_one(st.v[u].w, l > 3u ? sc : scb, l > 3u ? zp : zpb, qlo, qhi);
                } else {
                    st.v[u].x = fq_one(st.v[u].x, sc, zp, qlo, qhi);
                    st.v[u].y = fq_one(st.v[u].y, sc, zp, qlo, qhi);
                    st.v[u].z = fq_one(st.v[u].z, sc, zp, qlo, qhi);
                    st.v[u].w = fq_one(st.v[u].w, sc, zp, qlo, qhi);
                }
                __builtin_nontemporal_store(st.v[u], yv + i0 + u * kBlock);
            }
        }
    };
    if (tid < nvec) {
        Set A, B;
        uint32_t i0 = tid;
        load(A, i0);
        for (;;) {
            uint32_t nx = i0 + 4 * kBlock;
            const bool hb = nx < nvec;
            if (hb) load(B, nx);
            eat(A, i0);
            if (!hb) break;
            i0 = nx;
            nx = i0 + 4 * kBlock;
            const bool ha = nx < nvec;
            if (ha) load(A, nx);
            eat(B, i0);
            if (!ha) break;
            i0 = nx;
        }
    }
    const uint32_t t = (nvec << 2) + tid;   // (a chunk that is no multiple of four long: the tensor's last elements)
    if (t < cnt) {
        const uint32_t c = n_channels == 1u ? 0u : (uint32_t)(((e0 + t) / inner) % n_channels);
        ys[t] = fq_one(fq_pre<PRE>(xs[t], x2s[t]), scale_p[c], (float)zp_p[c], qlo, qhi);
    }
}

// one tensor: workgroup b takes elements [b * chunk, (b + 1) * chunk) (chunk a multiple of 1024)
// (PRE: the producer's ReLU / Add + ReLU on the way in, fq_pre; x2 is read for kFqPreAddRelu only)
template <int PRE>
__global__ __launch_bounds__(kBlock) void k_fake_quant(const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ y,
                                                        uint64_t n, uint64_t chunk, const float* __restrict__ scale_p,
                                                        const int32_t* __restrict__ zp_p, uint32_t n_channels, uint32_t inner, float qlo,
                                                        float qhi) {
    const uint64_t e0 = (uint64_t)blockIdx.x * chunk;
    if (e0 >= n) return;
    const uint64_t cnt = n - e0 < chunk ? n - e0 : chunk;
    fq_span<PRE>(x, x2, y, e0, (uint32_t)cnt, scale_p, zp_p, n_channels, inner, qlo, qhi);
}

// a whole tensor set in ONE launch: the balanced partition's items (item.seg = tensor, item.offset / count = the elements) over
// the tensors' base pointers and a parameter row per tensor
__global__ __launch_bounds__(kBlock) void k_fake_quant_items(const dpl_work_item* __restrict__ items, const uint32_t* __restrict__ bb,
                                                              const float* const* __restrict__ seg_x, float* const* __restrict__ seg_y,
                                                              const dpl_fake_quant_params* __restrict__ prm) {
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        const dpl_fake_quant_params p = prm[it.seg];
        fq_span<kFqPreNone>(seg_x[it.seg], nullptr, seg_y[it.seg], it.offset, it.count, p.d_scale, p.d_zero_point, (uint32_t)p.n_channels, (uint32_t)p.inner,
                (float)p.qlo, (float)p.qhi);
    }
}

// ================================================================ N1: cosine-similarity partial sums
__global__ __launch_bounds__(kBlock) void k_cos_acc(const float* __restrict__ a, const float* __restrict__ b,
                                                     int64_t n, double* __restrict__ acc) {
    __shared__ double s_r[3][kBlock / kWave];
    double ab = 0.0, aa = 0.0, bb = 0.0;
    const int64_t nvec = n >> 2;
    const f4* av = reinterpret_cast<const f4*>(a);
    const f4* bv = reinterpret_cast<const f4*>(b);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nvec; i0 += 4 * stride) {
        f4 p[4], q[4];   // eight 16-byte loads in flight per lane
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            p[u] = i < nvec ? __builtin_nontemporal_load(av + i) : f4{0.f, 0.f, 0.f, 0.f};
            q[u] = i < nvec ? __builtin_nontemporal_load(bv + i) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ab += (double)p[u].x * q[u].x + (double)p[u].y * q[u].y + (double)p[u].z * q[u].z + (double)p[u].w * q[u].w;
            aa += (double)p[u].x * p[u].x + (double)p[u].y * p[u].y + (double)p[u].z * p[u].z + (double)p[u].w * p[u].w;
            bb += (double)q[u].x * q[u].x + (double)q[u].y * q[u].y + (double)q[u].z * q[u].z + (double)q[u].w * q[u].w;
        }
    }
    const int64_t t = (nvec << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) {
        ab += (double)a[t] * b[t];
        aa += (double)a[t] * a[t];
        bb += (double)b[t] * b[t];
    }
    ab = wave_sum(ab);
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_r[0][w] = ab;
        s_r[1][w] = aa;
        s_r[2][w] = bb;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0.0;
        for (int k = 0; k < kBlock / kWave; ++k) v += s_r[threadIdx.x][k];
        atomicAdd(acc + threadIdx.x, v);
    }
}

// ================================================================ N2: per-channel sum of (a - b)  (bias correction)
// a, b viewed as [outer, C, inner] (Conv output [n, C, H, W]; Gemm output [n, C] with inner = 1):
// acc[c] += sum over outer and inner of (a - b), in fp64.  One wave per (outer, channel) row, rows round-robin over
// the waves of the launch; 16-byte loads when the rows allow it.
__global__ __launch_bounds__(kBlock) void k_channel_diff_sum(const float* __restrict__ a, const float* __restrict__ b,
                                                              uint64_t rows, uint32_t n_channels, uint32_t inner,
                                                              int vec_ok, double* __restrict__ acc) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
    if (inner == 1) {  // [n, C]: lanes over channels, waves over rows of 64 channels
        const uint64_t chunks = (n_channels + kWave - 1) / kWave;
        for (uint64_t t = wave; t < chunks; t += n_waves) {
            const uint32_t c = (uint32_t)t * kWave + lane;
            if (c >= n_channels) continue;
            double d = 0.0;
            for (uint64_t r = 0; r < rows / n_channels; ++r) d += (double)a[r * n_channels + c] - (double)b[r * n_channels + c];
            atomicAdd(acc + c, d);
        }
        return;
    }
    for (uint64_t r = wave; r < rows; r += n_waves) {
        const float* pa = a + r * inner;
        const float* pb = b + r * inner;
        double d = 0.0;
        uint32_t i = 0;
        if (vec_ok) {  // inner % 4 == 0 and both bases 16-byte aligned: every row starts aligned
            const f4* va = reinterpret_cast<const f4*>(pa);
            const f4* vb = reinterpret_cast<const f4*>(pb);
            const uint32_t nv = inner >> 2;
            for (uint32_t j = lane; j < nv; j += 2 * kWave) {
                const f4 p0 = __builtin_nontemporal_load(va + j), q0 = __builtin_nontemporal_load(vb + j);
                const bool two = j + kWave < nv;
                const f4 p1 = two ? __builtin_nontemporal_load(va + j + kWave) : f4{0.f, 0.f, 0.f, 0.f};
                const f4 q1 = two ? __builtin_nontemporal_load(vb + j + kWave) : f4{0.f, 0.f, 0.f, 0.f};
                d += ((double)p0.x - (double)q0.x) + ((double)p0.y - (double)q0.y) + ((double)p0.z - (double)q0.z) +
                     ((double)p0.w - (double)q0.w);
                d += ((double)p1.x - (double)q1.x) + ((double)p1.y - (double)q1.y) + ((double)p1.z - (double)q1.z) +
                     ((double)p1.w - (double)q1.w);
            }
            i = nv << 2;
        }
        for (uint32_t j = i + lane; j < inner; j += kWave) d += (double)pa[j] - (double)pb[j];
        d = wave_sum(d);
        if (lane == 0) atomicAdd(acc + (uint32_t)(r % n_channels), d);
    }
}

// Per-slot cosine partial sums over work items: slot = (image, tensor) pair for the profiling flow
// (profiling.py:57-64: one cosine per image per quantised layer output).  a and b come from two segment
// tables with identical geometry (fp model vs fake-quantised model).
__global__ __launch_bounds__(kBlock) void k_cos_items(const dpl_work_item* __restrict__ items,
                                                       const uint32_t* __restrict__ bb,
                                                       const float* const* __restrict__ segs_a,
                                                       const float* const* __restrict__ segs_b,
                                                       double* __restrict__ acc) {
    __shared__ double s_r[3][kBlock / kWave];
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        gptr_f32 a = (gptr_f32)(segs_a[it.seg] + it.offset);
        gptr_f32 b = (gptr_f32)(segs_b[it.seg] + it.offset);
        const uint32_t n = it.count;
        double ab = 0.0, aa = 0.0, bbs = 0.0;
        const bool vec = ((((uintptr_t)(segs_a[it.seg] + it.offset)) | ((uintptr_t)(segs_b[it.seg] + it.offset))) & 15u) == 0;
        uint32_t done = 0;
        if (vec) {
            const uint32_t nvec = n >> 2;
            gptr_f4 av = (gptr_f4)a;
            gptr_f4 bv = (gptr_f4)b;
            // two streams, software pipelined like stream_span: the next 2 + 2 vectors per lane are in flight while
            // the current ones are consumed (ping-pong register sets, no register copy between them)
            constexpr int kU = 2;
            constexpr uint32_t kStride = kU * kBlock;
            auto eat1 = [&](const f4& p, const f4& q) {
                ab += (double)p.x * q.x + (double)p.y * q.y + (double)p.z * q.z + (double)p.w * q.w;
                aa += (double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z + (double)p.w * p.w;
                bbs += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
            };
#define DPL_CLOAD(P, Q, base)                                      \
    _Pragma("unroll") for (int u = 0; u < kU; ++u) {               \
        P[u] = __builtin_nontemporal_load(av + (base) + u * kBlock); \
        Q[u] = __builtin_nontemporal_load(bv + (base) + u * kBlock); \
    }
#define DPL_CEAT(P, Q) _Pragma("unroll") for (int u = 0; u < kU; ++u) eat1(P[u], Q[u])
            uint32_t i = threadIdx.x;
            if (i + (kU - 1) * kBlock < nvec) {
                f4 PA[kU], QA[kU], PB[kU], QB[kU];
                DPL_CLOAD(PA, QA, i);
                i += kStride;
                for (;;) {
                    if (!(i + (kU - 1) * kBlock < nvec)) {
                        DPL_CEAT(PA, QA);
                        break;
                    }
                    DPL_CLOAD(PB, QB, i);
                    i += kStride;
                    DPL_CEAT(PA, QA);
                    if (!(i + (kU - 1) * kBlock < nvec)) {
                        DPL_CEAT(PB, QB);
                        break;
                    }
                    DPL_CLOAD(PA, QA, i);
                    i += kStride;
                    DPL_CEAT(PB, QB);
                }
            }
#undef DPL_CLOAD
#undef DPL_CEAT
            for (; i < nvec; i += kBlock) eat1(__builtin_nontemporal_load(av + i), __builtin_nontemporal_load(bv + i));
            done = nvec << 2;
        }
        for (uint32_t i = done + threadIdx.x; i < n; i += kBlock) {
            const float p = a[i], q = b[i];
            ab += (double)p * q;
            aa += (double)p * p;
            bbs += (double)q * q;
        }
        ab = wave_sum(ab);
        aa = wave_sum(aa);
        bbs = wave_sum(bbs);
        const int w = threadIdx.x / kWave;
        if ((threadIdx.x & (kWave - 1)) == 0) {
            s_r[0][w] = ab;
            s_r[1][w] = aa;
            s_r[2][w] = bbs;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double v = 0.0;
            for (int j = 0; j < kBlock / kWave; ++j) v += s_r[threadIdx.x][j];
            atomicAdd(acc + 3 * (uint64_t)it.slot + threadIdx.x, v);
        }
        __syncthreads();
    }
}

}  // namespace

// =================================================================================== C ABI
extern "C" {

int dpl_abi_version(void) { return DPL_ABI_VERSION; }
const char* dpl_last_error(void) { return g_err; }

int dpl_device_info(char* name, int name_cap, int* compute_units, uint64_t* hbm_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail("hipGetDevice", e);
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return fail("hipGetDeviceProperties", e);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) return fail_msg("current HIP device is not gfx950");
    return 0;
}

int dpl_stream_priority_range(int* least, int* greatest) {
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess) return fail("hipDeviceGetStreamPriorityRange", e);
    if (least) *least = lo;
    if (greatest) *greatest = hi;
    return 0;
}

int dpl_stream_create(int priority, dpl_stream_t* out) {
    if (!out) return fail_msg("dpl_stream_create: null out");
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);   // (lo: the numerically largest = least urgent)
    if (e != hipSuccess) return fail("hipDeviceGetStreamPriorityRange", e);
    if (priority > lo) priority = lo;
    if (priority < hi) priority = hi;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
    if (e != hipSuccess) return fail("hipStreamCreateWithPriority", e);
    *out = (dpl_stream_t)s;
    return 0;
}

int dpl_stream_destroy(dpl_stream_t s) {
    if (!s) return 0;
    hipError_t e = hipStreamDestroy((hipStream_t)s);
    return e == hipSuccess ? 0 : fail("hipStreamDestroy", e);
}

int64_t dpl_build_work_items(const dpl_span* spans, int64_t n_spans, uint64_t chunk_elems, dpl_work_item* out,
                             int64_t cap) {
    if (!spans || n_spans < 0 || chunk_elems == 0 || (chunk_elems % 1024) != 0 || chunk_elems > 0xFFFFFC00ull)
        return fail_msg("dpl_build_work_items: chunk_elems must be a non-zero multiple of 1024 below 2^32");
    int64_t n = 0;
    for (int64_t i = 0; i < n_spans; ++i) {
        uint64_t off = spans[i].offset, left = spans[i].count;
        while (left) {
            const uint64_t c = left < chunk_elems ? left : chunk_elems;
            if (out && n < cap) {
                out[n].offset = off;
                out[n].count = (uint32_t)c;
                out[n].seg = spans[i].seg;
                out[n].slot = spans[i].slot;
                out[n].reserved = 0;
            }
            ++n;
            off += c;
            left -= c;
        }
    }
    return n;
}

int dpl_minmax_init(uint32_t* d_min_enc, uint32_t* d_max_enc, uint32_t* d_nan, int64_t n_slots, dpl_stream_t s) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(k_minmax_init, dim3(grid_for(n_slots, 256)), dim3(256), 0, (hipStream_t)s, d_min_enc,
                       d_max_enc, d_nan, n_slots);
    DPL_LAUNCH_CHECK("k_minmax_init");
    return 0;
}

int64_t dpl_build_balanced_items(const dpl_span* spans, int64_t n_spans, int64_t n_blocks, dpl_work_item* out,
                                 int64_t cap, uint32_t* block_begin) {
    if (!spans || n_spans < 0 || n_blocks < 1) return fail_msg("dpl_build_balanced_items: bad arguments");
    unsigned __int128 total = 0;
    for (int64_t i = 0; i < n_spans; ++i) total += spans[i].count;
    int64_t n = 0;
    int64_t si = 0;
    uint64_t lo = 0;       // offset inside span si
    unsigned __int128 g = 0;  // global position of the cursor in the concatenated element stream
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (block_begin) block_begin[b] = (uint32_t)n;
        const unsigned __int128 target = (b + 1 == n_blocks) ? total : (total * (unsigned __int128)(b + 1)) / (unsigned __int128)n_blocks;
        while (si < n_spans && g < target) {
            const uint64_t remaining = spans[si].count - lo;
            if (remaining == 0) {
                ++si;
                lo = 0;
                continue;
            }
            const unsigned __int128 want = target - g;
            uint64_t take;
            bool span_done;
            if (want >= remaining) {
                take = remaining;
                span_done = true;
            } else {
                take = ((uint64_t)want / 1024u) * 1024u;  // cut points stay 4 KiB-aligned inside a span
                span_done = false;
                if (take == 0) break;  // less than one aligned piece left for this block: next block takes it
            }
            uint64_t off = spans[si].offset + lo, left = take;
            while (left) {  // a share larger than 2^32-1024 elements is emitted as several items
                const uint64_t c = left < 0xFFFFFC00ull ? left : 0xFFFFFC00ull;
                if (out && n < cap) {
                    out[n].offset = off;
                    out[n].count = (uint32_t)c;
                    out[n].seg = spans[si].seg;
                    out[n].slot = spans[si].slot;
                    out[n].reserved = 0;
                }
                ++n;
                off += c;
                left -= c;
            }
            g += take;
            lo += take;
            if (span_done) {
                ++si;
                lo = 0;
            } else {
                break;
            }
        }
    }
    if (block_begin) block_begin[n_blocks] = (uint32_t)n;
    return n;
}

// n_blocks = number of workgroups; d_block_begin (n_blocks + 1 entries) may be null, then n_blocks must equal
// n_items and workgroup b processes item b.
int dpl_minmax_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, uint32_t* d_min_enc,
                          uint32_t* d_max_enc, uint32_t* d_nan, dpl_stream_t s) {
    if (n_items <= 0) return 0;
    if (int e = check_blocks("dpl_minmax_accumulate", n_items, d_block_begin, n_blocks)) return e;
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)n_blocks), dim3(kBlock), 0, (hipStream_t)s, d_items, d_block_begin,
                       d_seg_ptrs, d_min_enc, d_max_enc, d_nan);
    DPL_LAUNCH_CHECK("k_minmax");
    return 0;
}

int dpl_minmax_finalize(const uint32_t* d_min_enc, const uint32_t* d_max_enc, const uint32_t* d_nan,
                        int64_t n_slots, float* d_min, float* d_max, dpl_stream_t s) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(k_minmax_finalize, dim3(grid_for(n_slots, 256)), dim3(256), 0, (hipStream_t)s, d_min_enc,
                       d_max_enc, d_nan, n_slots, d_min, d_max);
    DPL_LAUNCH_CHECK("k_minmax_finalize");
    return 0;
}

int dpl_minmax_encode(const float* d_min, const float* d_max, int64_t n_slots, uint32_t* d_min_enc,
                      uint32_t* d_max_enc, uint32_t* d_nan, dpl_stream_t s) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(k_minmax_encode, dim3(grid_for(n_slots, 256)), dim3(256), 0, (hipStream_t)s, d_min, d_max,
                       n_slots, d_min_enc, d_max_enc, d_nan);
    DPL_LAUNCH_CHECK("k_minmax_encode");
    return 0;
}

int dpl_hist_prepare(const float* d_min, const float* d_max, int64_t n_slots, int bins, dpl_hist_range* d_ranges,
                     dpl_stream_t s) {
    if (bins < 1 || bins > DPL_MAX_BINS) return fail_msg("dpl_hist_prepare: bins must be in [1, 16384]");
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(k_hist_prepare, dim3(grid_for(n_slots, 64)), dim3(64), 0, (hipStream_t)s, d_min, d_max,
                       n_slots, bins, d_ranges);
    DPL_LAUNCH_CHECK("k_hist_prepare");
    return 0;
}

int dpl_abs_hist_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                            int64_t n_blocks, const float* const* d_seg_ptrs, const dpl_hist_range* d_ranges,
                            int bins, uint64_t* d_hist, dpl_stream_t s) {
    if (bins < 1 || bins > DPL_MAX_BINS) return fail_msg("dpl_abs_hist_accumulate: bins must be in [1, 16384]");
    if (n_items <= 0) return 0;
    if (int e = check_blocks("dpl_abs_hist_accumulate", n_items, d_block_begin, n_blocks)) return e;
    hipLaunchKernelGGL(k_abs_hist, dim3((unsigned)n_blocks), dim3(kBlock),
                       ((size_t)bins + 1 + kBlock / kWave) * sizeof(uint32_t), (hipStream_t)s, d_items, d_block_begin,
                       d_seg_ptrs, d_ranges, bins, d_hist);
    DPL_LAUNCH_CHECK("k_abs_hist");
    return 0;
}

int dpl_hist_percentile(const uint64_t* d_hist, const float* d_min, const float* d_max, int64_t n_slots, int bins,
                        double threshold, float* d_clip, dpl_stream_t s) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(k_hist_percentile, dim3((unsigned)n_slots), dim3(kWave), 0, (hipStream_t)s, d_hist, d_min,
                       d_max, bins, threshold, d_clip);
    DPL_LAUNCH_CHECK("k_hist_percentile");
    return 0;
}

int dpl_rowwise_minmax(const float* d_w, int64_t rows, int64_t cols, float* d_min, float* d_max, dpl_stream_t s) {
    if (rows <= 0) return 0;
    if (cols <= 0 || cols > 0xFFFFFFFFll) return fail_msg("dpl_rowwise_minmax: cols out of range");
    hipLaunchKernelGGL(k_rowwise_minmax, dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)s, d_w, cols, d_min,
                       d_max);
    DPL_LAUNCH_CHECK("k_rowwise_minmax");
    return 0;
}

int dpl_fake_quant(const float* d_x, float* d_y, int64_t n, const float* d_scale, const int32_t* d_zp,
                   int64_t n_channels, int64_t inner, int32_t qlo, int32_t qhi, dpl_stream_t s) {
    return dpl_fake_quant_pre(DPL_FQ_PRE_NONE, d_x, nullptr, d_y, n, d_scale, d_zp, n_channels, inner, qlo, qhi, s);
}

int dpl_fake_quant_pre(int32_t pre, const float* d_x, const float* d_x2, float* d_y, int64_t n, const float* d_scale,
                       const int32_t* d_zp, int64_t n_channels, int64_t inner, int32_t qlo, int32_t qhi, dpl_stream_t s) {
    if (pre != DPL_FQ_PRE_NONE && pre != DPL_FQ_PRE_RELU && pre != DPL_FQ_PRE_ADD_RELU)
        return fail_msg("dpl_fake_quant_pre: pre must be DPL_FQ_PRE_NONE, _RELU or _ADD_RELU");
    if (n <= 0) return 0;
    if (pre == DPL_FQ_PRE_ADD_RELU && d_x2 == nullptr) return fail_msg("dpl_fake_quant_pre: DPL_FQ_PRE_ADD_RELU needs d_x2");
    if (n_channels < 1 || inner < 1 || n_channels > 0xFFFFFFFFll || inner > 0xFFFFFFFFll)
        return fail_msg("dpl_fake_quant_pre: n_channels and inner must be in [1, 2^32)");
    // A contiguous chunk of 3072 elements (12 KiB read + 12 KiB written) per workgroup, whatever the tensor's size (a multiple of
    // 1024 elements: every chunk starts on a 16-byte boundary of an aligned tensor).  Measured on the tensors a fake-quantised
    // ResNet-50 forward at batch 64 runs this on (26 - 205 MB, distinct buffers in rotation, scripts/fq_blocks_ab.py), fraction of
    // 8 TB/s by chunk: 1024: 0.61 / 0.52 (205 MB / 26 MB), 2048: 0.72 / 0.57, 3072: 0.76 / 0.56, 4096: 0.75 / 0.54, 8192: 0.78 /
    // 0.54, 12288: 0.72 / 0.43 — and rounds 3 - 4's rule (n / 4096 elements, at least 4096: 50 KB chunks for a 205 MB tensor):
    // 0.70 / 0.54.  The Q/DQ nodes of that forward: 0.61 -> 0.65 of the roofline (bench.py `fake_quant.product_forward`).
    // DPL_FQ_CHUNK: a tuning aid.
    static const int64_t chunk_elems = [] {
        const char* e = getenv("DPL_FQ_CHUNK");
        const int64_t v = e ? atoll(e) : 0;
        return v >= 1024 ? (v + 1023) / 1024 * 1024 : (int64_t)3072;
    }();
    int64_t chunk = chunk_elems;
    if ((n + chunk - 1) / chunk > 0x40000000ll) chunk = ((n + 0x3FFFFFFFll) / 0x40000000ll + 1023) / 1024 * 1024;
    if (chunk > 0xFFFFFC00ll) chunk = 0xFFFFFC00ll;
    const int64_t blocks = (n + chunk - 1) / chunk;
    if (blocks > 0x7FFFFFFFll) return fail_msg("dpl_fake_quant_pre: tensor too large");
#define DPL_FQ_LAUNCH(PRE)                                                                                                    \
    hipLaunchKernelGGL(k_fake_quant<PRE>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)s, d_x, d_x2, d_y, (uint64_t)n,   \
                       (uint64_t)chunk, d_scale, d_zp, (uint32_t)n_channels, (uint32_t)inner, (float)qlo, (float)qhi)
    if (pre == DPL_FQ_PRE_ADD_RELU) DPL_FQ_LAUNCH(kFqPreAddRelu);
    else if (pre == DPL_FQ_PRE_RELU) DPL_FQ_LAUNCH(kFqPreRelu);
    else DPL_FQ_LAUNCH(kFqPreNone);
#undef DPL_FQ_LAUNCH
    DPL_LAUNCH_CHECK("k_fake_quant");
    return 0;
}

int dpl_fake_quant_items(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                         const float* const* d_seg_x, float* const* d_seg_y, const dpl_fake_quant_params* d_params, dpl_stream_t s) {
    if (n_items <= 0) return 0;
    if (int e = check_blocks("dpl_fake_quant_items", n_items, d_block_begin, n_blocks)) return e;
    hipLaunchKernelGGL(k_fake_quant_items, dim3((unsigned)n_blocks), dim3(kBlock), 0, (hipStream_t)s, d_items, d_block_begin, d_seg_x,
                       d_seg_y, d_params);
    DPL_LAUNCH_CHECK("k_fake_quant_items");
    return 0;
}

int dpl_cos_accumulate(const float* d_a, const float* d_b, int64_t n, double* d_acc, int64_t slot, dpl_stream_t s) {
    if (n <= 0) return 0;
    if (((uintptr_t)d_a | (uintptr_t)d_b) & 15u) return fail_msg("dpl_cos_accumulate: buffers must be 16-B aligned");
    int64_t blocks = (n / 4 + kBlock * 8 - 1) / (kBlock * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cos_acc, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)s, d_a, d_b, n,
                       d_acc + 3 * slot);
    DPL_LAUNCH_CHECK("k_cos_acc");
    return 0;
}

int dpl_channel_diff_sum(const float* d_a, const float* d_b, int64_t outer, int64_t n_channels, int64_t inner,
                         double* d_acc, dpl_stream_t s) {
    if (outer <= 0 || n_channels <= 0 || inner <= 0) return 0;
    if (n_channels > 0xFFFFFFFFll || inner > 0xFFFFFFFFll) return fail_msg("dpl_channel_diff_sum: extent out of range");
    const uint64_t rows = (uint64_t)outer * (uint64_t)n_channels;
    const int vec_ok = ((inner & 3) == 0) && ((((uintptr_t)d_a | (uintptr_t)d_b) & 15u) == 0);
    uint64_t work = inner == 1 ? (uint64_t)(n_channels + kWave - 1) / kWave : rows;
    uint64_t blocks = (work + kBlock / kWave - 1) / (kBlock / kWave);
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_channel_diff_sum, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)s, d_a, d_b, rows,
                       (uint32_t)n_channels, (uint32_t)inner, vec_ok, d_acc);
    DPL_LAUNCH_CHECK("k_channel_diff_sum");
    return 0;
}

int dpl_cos_items_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                             int64_t n_blocks, const float* const* d_seg_a, const float* const* d_seg_b,
                             double* d_acc, dpl_stream_t s) {
    if (n_items <= 0) return 0;
    if (int e = check_blocks("dpl_cos_items_accumulate", n_items, d_block_begin, n_blocks)) return e;
    hipLaunchKernelGGL(k_cos_items, dim3((unsigned)n_blocks), dim3(kBlock), 0, (hipStream_t)s, d_items,
                       d_block_begin, d_seg_a, d_seg_b, d_acc);
    DPL_LAUNCH_CHECK("k_cos_items");
    return 0;
}

}  // extern "C"
