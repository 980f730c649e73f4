// OCTAV ('-A mse', forward_net.py:284-342) on MI355X: kernels + their C ABI entry points (include/dipoorlet_hip.h).
// Three forms of the same iterate sequence: full re-reads, tail compaction, and the two-read bracket form over an
// exact log-scale histogram (the default).  Shared helpers: common.hpp.
#include "common.hpp"
#include "octav_common.hpp"

#pragma clang fp contract(off)

namespace {

// ================================================================ K3: OCTAV (forward_net.py:315-330)
struct OctavFirstOp {
    float mn, mx;
    uint32_t nan, nz;
    double sum;
    __device__ __forceinline__ void operator()(float x) {
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
        nan |= (x != x);
        const float a = fabsf(x);
        nz += (a > 0.0f);
        sum += (double)a;
    }
};
struct OctavIterOp {
    float s;
    uint32_t gt, le;
    double sum;
    __device__ __forceinline__ void operator()(float x) {
        const float a = fabsf(x);
        const bool g = a > s;
        gt += g;
        le += (a <= s);
        sum += g ? (double)a : 0.0;
    }
};

template <bool kFirst>
__global__ __launch_bounds__(kBlock) void k_octav_pass(const dpl_work_item* __restrict__ items,
                                                        const uint32_t* __restrict__ bb,
                                                        const float* const* __restrict__ segs,
                                                        dpl_octav_state* __restrict__ st,
                                                        const dpl_octav_state* __restrict__ ctl) {
    // ctl (the extra state slot behind the pairs) counts the pairs in full-pass mode: nothing to do when 0
    if (!kFirst && ctl && ctl->cnt_gt == 0ull) return;
    __shared__ double s_sum[kBlock / kWave];
    __shared__ uint32_t s_a[kBlock / kWave], s_b[kBlock / kWave];
    __shared__ float s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    const int w = threadIdx.x / kWave;
    const bool lead = (threadIdx.x & (kWave - 1)) == 0;
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        dpl_octav_state* me = st + it.slot;
        const float* p = segs[it.seg] + it.offset;
        if (kFirst) {
            OctavFirstOp op{INFINITY, -INFINITY, 0u, 0u, 0.0};
            stream_span(p, it.count, op);
            const float mn = wave_min(op.mn), mx = wave_max(op.mx);
            const uint32_t nz = wave_sum(op.nz);
            const double sum = wave_sum(op.sum);
            const uint32_t nn = __any(op.nan) ? 1u : 0u;
            if (lead) {
                s_sum[w] = sum;
                s_a[w] = nz;
                s_b[w] = nn;
                s_mn[w] = mn;
                s_mx[w] = mx;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double tsum = 0.0;
                uint32_t tnz = 0, tnn = 0;
                float tmn = INFINITY, tmx = -INFINITY;
                for (int j = 0; j < kBlock / kWave; ++j) {
                    tsum += s_sum[j];
                    tnz += s_a[j];
                    tnn |= s_b[j];
                    tmn = fminf(tmn, s_mn[j]);
                    tmx = fmaxf(tmx, s_mx[j]);
                }
                atomicAdd(&me->sum, tsum);
                atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)tnz);
                atomicAdd(reinterpret_cast<unsigned long long*>(&me->n_elems), (unsigned long long)it.count);
                if (tmn <= tmx) {
                    atomicMin(&me->min_enc, enc_f32(tmn));
                    atomicMax(&me->max_enc, enc_f32(tmx));
                }
                if (tnn) atomicOr(&me->nan_seen, 1u);
            }
        } else {
            if (me->done || me->mode == 1u) continue;  // uniform per workgroup; list-mode pairs: k_octav_compact_*
            OctavIterOp op{me->s, 0u, 0u, 0.0};
            stream_span(p, it.count, op);
            const uint32_t gt = wave_sum(op.gt), le = wave_sum(op.le);
            const double sum = wave_sum(op.sum);
            if (lead) {
                s_sum[w] = sum;
                s_a[w] = gt;
                s_b[w] = le;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double tsum = 0.0;
                uint32_t tgt = 0, tle = 0;
                for (int j = 0; j < kBlock / kWave; ++j) {
                    tsum += s_sum[j];
                    tgt += s_a[j];
                    tle += s_b[j];
                }
                if (tgt) {
                    atomicAdd(&me->sum, tsum);
                    atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)tgt);
                }
                if (tle) atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_le), (unsigned long long)tle);
            }
        }
        __syncthreads();
    }
}

template <bool kFirst>
__global__ void k_octav_update(dpl_octav_state* __restrict__ st, int64_t n, int dynamic_sym, int max_iters,
                               dpl_octav_state* __restrict__ ctl) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dpl_octav_state* me = st + i;
    if (kFirst) {
        const float mn = dec_f32(me->min_enc);
        // forward_net.py:319 — np.abs(data_min - 0) < 1e-6 (float32 compare) and 'dynamic_sym' in qi_params
        me->unsigned_div = (dynamic_sym && fabsf(mn) < 1e-6f && !me->nan_seen) ? 4.0f : 1.0f;
        // forward_net.py:324 — s_n = abs_x.sum() / abs_x[abs_x > 0].size   (float32 / int)
        const float s0 = __fdiv_rn((float)me->sum, (float)(long long)me->cnt_gt);
        me->s = s0;
        me->iters = 0u;
        me->done = (s0 != s0 || max_iters <= 0) ? 1u : 0u;  // NaN is a fixed point of the iteration
    } else {
        if (me->done || me->mode == 2u) return;
        // list mode evaluates only the tail: everything not above s is below or equal (no NaN: those pairs are done)
        const unsigned long long cnt_le = me->mode == 1u ? me->n_elems - me->cnt_gt : me->cnt_le;
        const float s_before = me->s;
        const OctavStep r = octav_step(me->sum, me->cnt_gt, cnt_le, me->unsigned_div, me->s, me->iters, max_iters);
        me->s = r.s;
        me->iters = r.iters;
        me->done = r.done;
        // the list just written holds the values above the old s: it cannot answer for a smaller threshold
        if (me->mode == 1u && r.decreased && !r.done) {
            me->mode = 0u;
            if (ctl) atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_gt), 1ull);
        }
        if (me->mode == 1u) {
            me->cur = (me->cur == 2u) ? 0u : 1u - me->cur;  // the freshly written list is the next source
            me->len[1u - me->cur] = 0u;                       // ... and the other one the next destination
            me->reserved = __float_as_uint(s_before);         // ... which holds the values above the iterate it was built at
        }
    }
    me->sum = 0.0;
    me->cnt_gt = 0ull;
    me->cnt_le = 0ull;
}

// ---------------------------------------------------------------- OCTAV with tail compaction
// The iterates climb (s_{k+1} >= s_k while below the fixed point), so evaluation k only needs the values
// above s_{k-1}.  The first evaluation reads the full data once and writes the values above s_0; each later
// one reads the previous list and writes the next, and the lists shrink ~2.5x per step.  Exactly the same
// iterate sequence as the full-pass form; a pair whose iterate ever decreases drops back to full passes.
constexpr int kStageCap = 2048;  // floats of LDS staging per wave

struct TailAcc {
    uint32_t gt;  // wave-uniform: survivors this wave has seen
    double sum;   // per lane
};

// One wave, one 1024-element tile in registers: survivors (|x| > s) go to the wave's LDS stage.
// Per element column j: the wave's ballot gives every surviving lane its slot (v_mbcnt) and the stage
// cursor advances by the population count on the scalar unit — no cross-lane scan, conflict-free writes.
// Values are summed in fp32 over the lane's 16 elements, then added to the fp64 accumulator (at least as
// accurate as numpy's blocked fp32 pairwise sum).
template <int kCap = kStageCap, class FlushFn>
__device__ __forceinline__ void tail_tile(const f4 (&v)[4], float s, float* stage, uint32_t& fill, TailAcc& acc,
                                          FlushFn&& flush) {
    if (fill + 1024u > (uint32_t)kCap) flush();  // wave-uniform
    float a[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        a[4 * u + 0] = fabsf(v[u].x);
        a[4 * u + 1] = fabsf(v[u].y);
        a[4 * u + 2] = fabsf(v[u].z);
        a[4 * u + 3] = fabsf(v[u].w);
    }
    float part = 0.0f;
    const uint32_t fill0 = fill;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool g = a[j] > s;
        const unsigned long long m = __ballot(g);
        const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (g) stage[fill + off] = a[j];
        fill += (uint32_t)__popcll(m);
        part += g ? a[j] : 0.0f;
    }
    acc.gt += fill - fill0;
    acc.sum += (double)part;
}

// First evaluation: full data -> list 0, several workgroups per pair (global cursor + atomics).
__global__ __launch_bounds__(kBlock) void k_octav_compact_full(const dpl_work_item* __restrict__ items,
                                                                const uint32_t* __restrict__ bb,
                                                                const float* const* __restrict__ segs,
                                                                dpl_octav_state* __restrict__ st,
                                                                const dpl_octav_state* __restrict__ ctl,
                                                                const uint64_t* __restrict__ pair_base,
                                                                float* __restrict__ list0) {
    extern __shared__ __attribute__((aligned(16))) float stage_all[];
    __shared__ double s_sum[kBlock / kWave];
    __shared__ uint32_t s_gt[kBlock / kWave];
    const int w = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    float* stage = stage_all + w * kStageCap;
    if (ctl && ctl->cnt_le == 0ull) return;  // no pair on the compaction route
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        dpl_octav_state* me = st + it.slot;
        if (me->done || me->mode != 1u) continue;
        const float s = me->s;
        const float* p = segs[it.seg] + it.offset;
        float* dst = list0 + pair_base[it.slot];
        uint32_t fill = 0;
        TailAcc acc{0u, 0.0};
        auto flush = [&]() {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&me->len[0], fill);
            base = __shfl(base, 0, kWave);
            for (uint32_t j = lane; j < fill; j += kWave) dst[base + j] = stage[j];
            fill = 0;
        };
        for_each_tile<kBlock>(p, it.count, [&](const f4 (&v)[4], uint32_t, bool) { tail_tile(v, s, stage, fill, acc, flush); });
        if (fill) flush();
        const uint32_t gt = acc.gt;  // already wave-uniform
        const double sum = wave_sum(acc.sum);
        if (lane == 0) {
            s_gt[w] = gt;
            s_sum[w] = sum;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tg = 0;
            double ts = 0.0;
            for (int j = 0; j < kBlock / kWave; ++j) {
                tg += s_gt[j];
                ts += s_sum[j];
            }
            if (tg) {
                atomicAdd(&me->sum, ts);
                atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)tg);
            }
        }
        __syncthreads();
    }
}

// Later evaluations: ONE persistent workgroup per pair walks the remaining iterations by itself — evaluate
// at s over list[cur], compact the survivors into list[1 - cur], take the fixed-point step, swap — with no
// kernel boundary in between (the lists shrink ~2.5x per step and stay in this XCD's L2).
constexpr int kIterBlock = 512;      // 8 waves per pair; 64 KiB of LDS staging -> 2 workgroups per CU
constexpr int kIterStageCap = 2048;  // floats of LDS staging per wave (two full tiles)

__global__ __launch_bounds__(kIterBlock) void k_octav_iterate_lists(dpl_octav_state* __restrict__ st,
                                                                     dpl_octav_state* __restrict__ ctl,
                                                                     const uint32_t* __restrict__ pair_order,
                                                                     const uint64_t* __restrict__ pair_base,
                                                                     float* __restrict__ list0,
                                                                     float* __restrict__ list1, int max_iters) {
    extern __shared__ __attribute__((aligned(16))) float stage_all[];
    constexpr int kWaves = kIterBlock / kWave;
    __shared__ double s_sum[kWaves];
    __shared__ uint32_t s_gt[kWaves];
    __shared__ uint32_t s_cursor;
    __shared__ OctavStep s_step;
    // largest pairs first (pair_order is sorted by size): the long sequential chains start at once and the
    // short ones fill the tail of the launch
    if (ctl->cnt_le == 0ull) return;  // no pair on the compaction route
    const uint32_t pair = pair_order ? pair_order[blockIdx.x] : blockIdx.x;
    dpl_octav_state* me = st + pair;
    if (me->done || me->mode != 1u || me->cur > 1u) return;  // uniform per workgroup
    const int w = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    float* stage = stage_all + w * kIterStageCap;
    const uint64_t base_off = pair_base[pair];
    const unsigned long long n_elems = me->n_elems;
    const float unsigned_div = me->unsigned_div;
    float s = me->s;
    uint32_t iters = me->iters, cur = me->cur, n = me->len[cur];
    uint32_t done = 0u, decreased = 0u;
    // list[cur] holds exactly the values above the iterate it was produced at (kept by k_octav_update in `reserved`)
    float s_floor_l = __uint_as_float(me->reserved);
    while (!done && !decreased) {
        const float* src = (cur == 0 ? list0 : list1) + base_off;
        float* dst = (cur == 0 ? list1 : list0) + base_off;
        if (n <= (uint32_t)(kIterBlock * 32)) {
            // the tail now fits the workgroup's registers (32 values per lane): finish every remaining
            // iteration without touching memory again — each one is a compare, a reduction and a step
            gptr_f32 g = (gptr_f32)src;
            float r[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const uint32_t idx = j * kIterBlock + threadIdx.x;
                r[j] = idx < n ? g[idx] : 0.0f;  // zeros never exceed s >= 0
            }
            const float floor_s = s_floor_l;  // every value above this is in the registers
            while (!done && !decreased) {
                uint32_t c = 0;
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const bool g0 = r[j] > s, g1 = r[j + 1] > s;
                    c += (uint32_t)g0 + (uint32_t)g1;
                    p0 += g0 ? r[j] : 0.0f;
                    p1 += g1 ? r[j + 1] : 0.0f;
                }
                c = wave_sum(c);
                double sm = wave_sum((double)p0 + (double)p1);
                if (lane == 0) {
                    s_gt[w] = c;
                    s_sum[w] = sm;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    unsigned long long tg = 0;
                    double ts = 0.0;
                    for (int j = 0; j < kWaves; ++j) {
                        tg += s_gt[j];
                        ts += s_sum[j];
                    }
                    s_step = octav_step(ts, tg, n_elems - tg, unsigned_div, s, iters, max_iters);
                }
                __syncthreads();
                const OctavStep st2 = s_step;
                s = st2.s;
                iters = st2.iters;
                done = st2.done;
                decreased = st2.decreased && !(st2.s >= floor_s);  // a dip is fine while nothing needed was dropped
                __syncthreads();
            }
            break;
        }
        if (threadIdx.x == 0) s_cursor = 0u;
        __syncthreads();
        uint32_t fill = 0;
        TailAcc acc{0u, 0.0};
        auto flush = [&]() {
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(&s_cursor, fill);
            b0 = __shfl(b0, 0, kWave);
            for (uint32_t j = lane; j < fill; j += kWave) dst[b0 + j] = stage[j];
            fill = 0;
        };
        for_each_tile<kIterBlock>(src, n, [&](const f4 (&v)[4], uint32_t, bool) {
            tail_tile<kIterStageCap>(v, s, stage, fill, acc, flush);
        });
        if (fill) flush();
        const uint32_t gt = acc.gt;  // already wave-uniform
        const double sum = wave_sum(acc.sum);
        if (lane == 0) {
            s_gt[w] = gt;
            s_sum[w] = sum;
        }
        __syncthreads();  // also orders every wave's dst stores before the next round reads them
        if (threadIdx.x == 0) {
            unsigned long long tg = 0;
            double ts = 0.0;
            for (int j = 0; j < kWaves; ++j) {
                tg += s_gt[j];
                ts += s_sum[j];
            }
            s_step = octav_step(ts, tg, n_elems - tg, unsigned_div, s, iters, max_iters);
        }
        __threadfence_block();
        __syncthreads();
        const OctavStep r = s_step;
        n = s_cursor;
        s_floor_l = s;  // the list just written holds the values above the iterate it was evaluated at
        s = r.s;
        iters = r.iters;
        done = r.done;
        decreased = r.decreased;
        cur = 1u - cur;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        me->s = s;
        me->iters = iters;
        me->done = done;
        me->cur = cur;
        me->len[cur] = n;
        me->len[1u - cur] = 0u;
        me->sum = 0.0;
        me->cnt_gt = 0ull;
        me->cnt_le = 0ull;
        if (!done && decreased) {  // the iterate went down: the tail list cannot answer; finish on the full data
            me->mode = 0u;
            atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_gt), 1ull);
        }
    }
}

// Fallback for the (degenerate) pairs that left list mode: one workgroup per pair finishes the iteration on
// the pair's full data.  Returns at once when the control block counts no such pair.
__global__ __launch_bounds__(kBlock) void k_octav_iterate_full(dpl_octav_state* __restrict__ st,
                                                                const dpl_octav_state* __restrict__ ctl,
                                                                const dpl_span* __restrict__ pair_spans,
                                                                const float* const* __restrict__ segs, int max_iters) {
    if (ctl->cnt_gt == 0ull) return;
    __shared__ double s_sum[kBlock / kWave];
    __shared__ uint32_t s_a[kBlock / kWave], s_b[kBlock / kWave];
    __shared__ OctavStep s_step;
    dpl_octav_state* me = st + blockIdx.x;
    if (me->done || me->mode != 0u) return;
    const dpl_span sp = pair_spans[blockIdx.x];
    const float* p = segs[sp.seg] + sp.offset;
    const int w = threadIdx.x / kWave;
    const bool lead = (threadIdx.x & (kWave - 1)) == 0;
    float s = me->s;
    uint32_t iters = me->iters, done = 0u;
    const float unsigned_div = me->unsigned_div;
    while (!done) {
        OctavIterOp op{s, 0u, 0u, 0.0};
        stream_span(p, (uint32_t)sp.count, op);
        const uint32_t gt = wave_sum(op.gt), le = wave_sum(op.le);
        const double sum = wave_sum(op.sum);
        if (lead) {
            s_sum[w] = sum;
            s_a[w] = gt;
            s_b[w] = le;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ts = 0.0;
            unsigned long long tg = 0, tl = 0;
            for (int j = 0; j < kBlock / kWave; ++j) {
                ts += s_sum[j];
                tg += s_a[j];
                tl += s_b[j];
            }
            s_step = octav_step(ts, tg, tl, unsigned_div, s, iters, max_iters);
        }
        __syncthreads();
        const OctavStep r = s_step;
        s = r.s;
        iters = r.iters;
        done = r.done;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        me->s = s;
        me->iters = iters;
        me->done = 1u;
    }
}

__global__ void k_octav_init(dpl_octav_state* st, int64_t n, uint32_t mode) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;  // slot n is the control block
    dpl_octav_state z;
    z.sum = 0.0;
    z.cnt_gt = 0;
    z.cnt_le = 0;
    z.min_enc = 0xFFFFFFFFu;
    z.max_enc = 0u;
    z.nan_seen = 0u;
    z.done = 0u;
    z.s = 0.0f;
    z.unsigned_div = 1.0f;
    z.iters = 0u;
    z.mode = mode;
    z.n_elems = 0ull;
    z.len[0] = 0u;
    z.len[1] = 0u;
    z.cur = 2u;
    z.reserved = 0u;
    if (i == n) {  // control block: cnt_gt = pairs in full-pass mode, cnt_le = pairs on the compaction route
        z.cnt_gt = mode == 0u ? (unsigned long long)n : 0ull;
        z.cnt_le = mode == 1u ? (unsigned long long)n : 0ull;
    }
    st[i] = z;
}

__global__ void k_octav_finalize(const dpl_octav_state* st, int64_t n, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool bad = st[i].nan_seen != 0u || st[i].min_enc == 0xFFFFFFFFu;
    // control block bit 1: a workgroup of the resident form gave up waiting for its cluster (never expected)
    out[3 * i + 0] = (st[n].nan_seen & 2u) ? __uint_as_float(0x7FC00000u) : st[i].s;
    out[3 * i + 1] = bad ? NAN : dec_f32(st[i].min_enc);
    out[3 * i + 2] = bad ? NAN : dec_f32(st[i].max_enc);
}


// ================================================================ K3c: OCTAV through a log-scale histogram
// Goal: two reads of the data, no tail lists.  Pass 1 (with the statistics) bins |x| by its float bit
// pattern — 64 sub-bins per octave over 2^-18 .. 2^14, i.e. bin = (bits >> 17) - key0 — keeping per bin an
// exact count and an exact integer sum of mantissas (all values of a bin share the exponent, so
// sum = (sum of 24-bit mantissas) * 2^(e-150): order-independent, deterministic).  F(s) is then exact at every
// bin edge.  A small per-pair kernel walks the iteration in BRACKET form over the edges and marks the few
// dozen bins the true iterates can fall into (about 2 % of the elements); pass 2 gathers just those elements; a
// per-pair kernel then runs the reference's exact iteration from (exact bin totals above the current bin) +
// (gathered elements of the current bin).  Every iterate is verified to land in a marked bin; a pair that
// fails (or whose bracket explodes: flat / degenerate distributions) takes the compaction path instead.
struct LogHistOp {
    unsigned long long* packed;
    float mn, mx;
    uint32_t nan;         // only ever examined on the rare path below
    uint32_t nz_out;      // nonzero values outside the binned window (|x| < 2^-18 or >= 2^14)
    double sum_out;
    __device__ __forceinline__ void operator()(float x) {
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
        // window bins 1 .. kLogNB-1 (bin 0 = everything below 2^-18 is never needed: counts below an iterate come
        // from n_elems).  In-window values are all positive and finite, so the pair's sum(|x|) and count(|x| > 0)
        // follow from the histogram itself; only what falls outside (zeros, denormal-small, huge, inf, NaN) takes
        // the second branch, and only its nonzero members (rare) are accumulated directly.
        const uint32_t bits = __float_as_uint(x);
        const uint32_t t = ((bits >> kLogShift) & 0x3FFFu) - (kLogKey0 + 1u);
        if (t < (uint32_t)(kLogNB - 1)) {
            // the packed word sums the 23 explicit mantissa bits; the implicit ones are count << 23 (flush)
            atomicAdd(packed + t + 1u, (1ull << kPackShift) | (unsigned long long)(bits & 0x7FFFFFu));
        } else if (__any(!(fabsf(x) <= 0.0f))) {     // wave-uniform: zeros alone skip the block
            const float a = fabsf(x);
            if (a > 0.0f) {
                sum_out += (double)a;
                ++nz_out;
            }
            nan |= (a != a);
        }
    }
};

__global__ __launch_bounds__(kBlock) void k_octav_loghist(const dpl_work_item* __restrict__ items,
                                                           const uint32_t* __restrict__ bb,
                                                           const float* const* __restrict__ segs,
                                                           dpl_octav_state* __restrict__ st,
                                                           uint32_t* __restrict__ lh_cnt,
                                                           unsigned long long* __restrict__ lh_sum) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long l_packed[];  // kLogNB packed bins
    __shared__ double s_sum[kBlock / kWave];
    __shared__ uint32_t s_a[kBlock / kWave], s_b[kBlock / kWave];
    __shared__ float s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    const int w = threadIdx.x / kWave;
    const bool lead = (threadIdx.x & (kWave - 1)) == 0;
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        dpl_octav_state* me = st + it.slot;
        for (int b = threadIdx.x; b < kLogNB; b += kBlock) l_packed[b] = 0ull;
        __syncthreads();
        LogHistOp op{l_packed, INFINITY, -INFINITY, 0u, 0u, 0.0};  // mn, mx, nan, nz_out, sum_out
        uint32_t* gc = lh_cnt + (uint64_t)it.slot * kLogNB;
        unsigned long long* gs = lh_sum + (uint64_t)it.slot * kLogNB;
        // sub-spans below 2^20 elements keep the packed count field from overflowing
        constexpr uint32_t kSub = (1u << 20) - 4096u;
        for (uint32_t s0 = 0; s0 < it.count; s0 += kSub) {
            stream_span(segs[it.seg] + it.offset + s0, min(kSub, it.count - s0), op);
            __syncthreads();
            for (int b = threadIdx.x; b < kLogNB; b += kBlock) {
                const unsigned long long v = l_packed[b];
                if (v) {
                    const unsigned long long c = v >> kPackShift;
                    atomicAdd(gc + b, (uint32_t)c);
                    atomicAdd(gs + b, (v & kPackMask) + (c << 23));   // full 24-bit mantissas
                    l_packed[b] = 0ull;
                }
            }
            __syncthreads();
        }
        const float mn = wave_min(op.mn), mx = wave_max(op.mx);
        const uint32_t nz = wave_sum(op.nz_out);
        const double sum = wave_sum(op.sum_out);
        const uint32_t nn = __any(op.nan) ? 1u : 0u;
        if (lead) {
            s_sum[w] = sum;
            s_a[w] = nz;
            s_b[w] = nn;
            s_mn[w] = mn;
            s_mx[w] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double tsum = 0.0;
            uint32_t tnz = 0, tnn = 0;
            float tmn = INFINITY, tmx = -INFINITY;
            for (int j = 0; j < kBlock / kWave; ++j) {
                tsum += s_sum[j];
                tnz += s_a[j];
                tnn |= s_b[j];
                tmn = fminf(tmn, s_mn[j]);
                tmx = fmaxf(tmx, s_mx[j]);
            }
            if (tnz) {  // out-of-window part only: the bracket kernel adds the histogram totals
                atomicAdd(&me->sum, tsum);
                atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)tnz);
            }
            atomicAdd(reinterpret_cast<unsigned long long*>(&me->n_elems), (unsigned long long)it.count);
            if (tmn <= tmx) {
                atomicMin(&me->min_enc, enc_f32(tmn));
                atomicMax(&me->max_enc, enc_f32(tmx));
            }
            if (tnn) atomicOr(&me->nan_seen, 1u);
        }
        __syncthreads();
    }
}

// Shared by the bracket walk and the exact walk: suffix totals over the bins, S_ge[j] / N_ge[j] = everything
// in bins >= j.  Built by one workgroup per pair into LDS (N as u32, S as fp64; j = 0 .. kLogNB).
__device__ __forceinline__ void build_suffix(uint32_t* gc, unsigned long long* gs, uint32_t* n_ge, double* s_ge,
                                             double* scratch_s, uint32_t* scratch_n, bool write_back) {
    // each thread owns a run of consecutive bins (thread 0 the top ones), read once into registers; exclusive
    // prefix over threads by a wave-level shuffle scan + a serial pass over the wave totals  (kBlock threads)
    constexpr int kPerT = kLogNB / kBlock;
    static_assert(kLogNB % kBlock == 0, "bins must split evenly over the workgroup");
    const int hi = kLogNB - 1 - (int)threadIdx.x * kPerT;  // my highest bin
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    uint32_t cn[kPerT];
    double cs[kPerT];
    double ls = 0.0;
    uint32_t ln = 0;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        cn[q] = gc[hi - q];
        cs[q] = (double)gs[hi - q] * log_bin_scale(hi - q);
    }
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        ln += cn[q];
        ls += cs[q];
    }
    double is = ls;
    uint32_t in = ln;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double ts = __shfl_up(is, o, kWave);
        const uint32_t tn = __shfl_up(in, o, kWave);
        if (lane >= (uint32_t)o) {
            is += ts;
            in += tn;
        }
    }
    if (lane == kWave - 1) {
        scratch_s[w] = is;
        scratch_n[w] = in;
    }
    __syncthreads();
    double rs = is - ls;  // exclusive within the wave
    uint32_t rn = in - ln;
    for (int q = 0; q < w; ++q) {
        rs += scratch_s[q];
        rn += scratch_n[q];
    }
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        const int bq = hi - q;
        rn += cn[q];
        rs += cs[q];
        n_ge[bq] = rn;
        s_ge[bq] = rs;
        if (write_back) {  // each thread owns its bins: the raw histogram is replaced by the suffix totals
            gc[bq] = rn;
            gs[bq] = (unsigned long long)__double_as_longlong(rs);
        }
    }
    if (threadIdx.x == 0) {
        n_ge[kLogNB] = 0u;
        s_ge[kLogNB] = 0.0;
    }
    __syncthreads();
}

// Per pair: s_0, then the bracket walk over the bin edges; marks the bins the iterates can visit.
__global__ __launch_bounds__(kBlock) void k_octav_bracket(dpl_octav_state* __restrict__ st,
                                                           dpl_octav_state* __restrict__ ctl,
                                                           uint32_t* __restrict__ lh_cnt,
                                                           unsigned long long* __restrict__ lh_sum,
                                                           uint32_t* __restrict__ bitmap, int dynamic_sym, int max_iters) {
    __shared__ uint32_t n_ge[kLogNB + 1];
    __shared__ double s_ge[kLogNB + 1];
    __shared__ double scr_s[kBlock];
    __shared__ uint32_t scr_n[kBlock];
    __shared__ uint32_t bm[kLogWords];
    __shared__ uint32_t route;  // 0: done already, 2: bracket route, 1: compaction route
    __shared__ int jmin_s, jmax_s;
    if (threadIdx.x == 0) {
        jmin_s = kLogNB;
        jmax_s = -1;
    }
    const int64_t pr = blockIdx.x;
    dpl_octav_state* me = st + pr;
    uint32_t* gc = lh_cnt + pr * kLogNB;
    unsigned long long* gs = lh_sum + pr * kLogNB;
    if (threadIdx.x < kLogWords) bm[threadIdx.x] = 0u;
    // the exact walk (k_octav_exact) needs the totals above a bin, never a single bin: from here on the pair's
    // histogram rows hold the suffix totals  N_ge[j] (u32)  and  S_ge[j] (fp64 bits)
    build_suffix(gc, gs, n_ge, s_ge, scr_s, scr_n, true);
    if (threadIdx.x == 0) {
        const BracketResult br = bracket_walk(n_ge, s_ge, bm, dec_f32(me->min_enc), dec_f32(me->max_enc), me->nan_seen != 0u,
                                              me->sum, me->cnt_gt, me->n_elems, dynamic_sym, max_iters);
        me->unsigned_div = br.unsigned_div;
        me->s = br.s0;
        me->iters = 0u;
        me->sum = 0.0;
        me->cnt_gt = 0ull;
        me->cnt_le = 0ull;
        me->len[0] = 0u;
        me->len[1] = 0u;
        me->cur = 2u;
        jmin_s = br.jmin;
        jmax_s = br.jmax;
        if (br.route == 0u) me->done = 1u;  // NaN is a fixed point of the iteration
        if (br.route == 1u) {  // compaction route (k_octav_compact_full and friends)
            me->mode = 1u;
            atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
        } else {
            me->mode = 2u;
        }
        route = br.route;
    }
    __syncthreads();
    // bitmap row: kLogWords words of marks + [lowest marked edge, edge above the highest marked bin] as float bits
    if (threadIdx.x < kLogWords) bitmap[pr * kBitmapRow + threadIdx.x] = (route == 2u) ? bm[threadIdx.x] : 0u;
    if (threadIdx.x == 0) {
        const int jmin = route == 2u ? jmin_s : kLogNB, jmax = route == 2u ? jmax_s : -1;
        const float rlo = jmax < 0 ? INFINITY : log_edge(jmin);
        const float rhi = jmax < 0 ? -INFINITY : (jmax >= kLogNB - 1 ? INFINITY : log_edge(jmax + 1));
        bitmap[pr * kBitmapRow + kLogWords] = __float_as_uint(rlo);
        bitmap[pr * kBitmapRow + kLogWords + 1] = __float_as_uint(rhi);
    }
}

// Pass 2: collect the elements that fall in marked bins (|x| values) into the pair's list 0.  Survivors are
// sparse (about 2 %), so each lane appends to a private LDS queue (no cross-lane work per element); a wave flushes
// its queues behind one scan + one returning atomic when any queue is half full.
// The membership test is one LDS word + a bit extract per element: the pair's 2048 window marks are placed inside a
// bitmap over the WHOLE key space (bits >> 17 of any non-negative float: 16384 keys = 2 KiB), so no clamping or
// rebasing is needed per element and zeros / padding / out-of-window values fall on words that are never set.
// The append is branch-free: every element is written at the lane's queue tail and the tail only advances for a
// survivor (2 VALU + 1 LDS write per element instead of a predicated block per element).
#ifndef DPL_QUEUE_CAP
#define DPL_QUEUE_CAP 32
#endif
constexpr int kQueueCap = DPL_QUEUE_CAP;     // a tile adds at most 16: flush once a queue holds more than cap - 16
constexpr int kQueueStride = kQueueCap + 1;  // entries per lane (+1: the branch-free append writes one past the fill)
constexpr int kKeyWords = (1 << (31 - kLogShift)) / 32;   // 512
constexpr int kKeyWord0 = (int)(kLogKey0 >> 5);           // word of the window's first bin (kLogKey0 is a multiple of 32)
static_assert((kLogKey0 & 31u) == 0u, "the window must start on a bitmap word");

// One span of one pair: the values of the marked bins (bm: the bitmap over the whole key space, in LDS) -> per-lane queues
// -> the pair's list (cursor: me->len[0], a returning global atomic per wave flush).
// cap: values the destination region holds — a flush that would pass it is dropped while the cursor keeps counting, so a
// length above the region's capacity tells the reader that the list is incomplete (the rescue walk then refuses the pair).
__device__ __forceinline__ void gather_span(const float* __restrict__ p, uint32_t count, const uint32_t* bm, uint32_t* q,
                                            dpl_octav_state* me, uint32_t* __restrict__ dst, uint32_t cap = 0xFFFFFFFFu) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    uint32_t cnt = 0;
    auto flush = [&]() {
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, kWave);
            if (lane >= (uint32_t)o) inc += t;
        }
        const uint32_t total = __shfl(inc, kWave - 1, kWave);
        uint32_t base = 0;
        if (lane == kWave - 1) base = atomicAdd(&me->len[0], total);
        const uint32_t first = __shfl(base, kWave - 1, kWave);
        base = first + inc - cnt;
        if (first + total <= cap)
            for (uint32_t j = 0; j < cnt; ++j) dst[base + j] = q[j * kWave];
        cnt = 0;
    };
    for_each_tile<kBlock>(p, count, [&](const f4 (&v)[4], uint32_t, bool) {
        // phase 1: all 16 bitmap words are fetched back to back (one LDS wait for the tile)
        uint32_t u[16], word[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u[4 * c + 0] = __float_as_uint(v[c].x) & 0x7FFFFFFFu;
            u[4 * c + 1] = __float_as_uint(v[c].y) & 0x7FFFFFFFu;
            u[4 * c + 2] = __float_as_uint(v[c].z) & 0x7FFFFFFFu;
            u[4 * c + 3] = __float_as_uint(v[c].w) & 0x7FFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) word[j] = bm[u[j] >> (kLogShift + 5)];
        uint32_t hit[16], any = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            hit[j] = (word[j] >> ((u[j] >> kLogShift) & 31u)) & 1u;
            any |= hit[j];
        }
        // phase 2: branch-free append (a tile without any survivor in the wave skips it)
        if (__any(any != 0u)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                q[cnt * kWave] = u[j];
                cnt += hit[j];
            }
        }
        if (__any(cnt > (uint32_t)(kQueueCap - 16))) flush();
    });
    if (__any(cnt != 0u)) flush();
}

__global__ __launch_bounds__(kBlock) void k_octav_gather(const dpl_work_item* __restrict__ items,
                                                          const uint32_t* __restrict__ bb,
                                                          const float* const* __restrict__ segs,
                                                          dpl_octav_state* __restrict__ st,
                                                          const uint32_t* __restrict__ bitmap,
                                                          const uint64_t* __restrict__ pair_base,
                                                          float* __restrict__ list0) {
    extern __shared__ __attribute__((aligned(16))) uint32_t queues[];  // [waves][kQueueStride][64]: entry j of lane l
    // sits at [j][l], so the bank of every queue access depends on the lane alone (no conflicts whatever the fills)
    __shared__ uint32_t bm[kKeyWords];
    const int w = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    uint32_t* q = queues + (size_t)w * kWave * kQueueStride + lane;
    uint32_t k0, k1;
    block_items(bb, k0, k1);
    for (uint32_t k = k0; k < k1; ++k) {
        const dpl_work_item it = items[k];
        dpl_octav_state* me = st + it.slot;
        if (me->done || me->mode != 2u) continue;  // uniform per workgroup
        __syncthreads();
        for (int i = threadIdx.x; i < kKeyWords; i += kBlock) {
            const int j = i - kKeyWord0;
            bm[i] = (j >= 0 && j < kLogWords) ? bitmap[(uint64_t)it.slot * kBitmapRow + j] : 0u;
        }
        __syncthreads();
        gather_span(segs[it.seg] + it.offset, it.count, bm, q, me,
                    reinterpret_cast<uint32_t*>(list0 + pair_base[it.slot]));
    }
}

// The RESCUE of the one-read form (octav_tail_host.hip): the pairs whose walk stepped outside the gathered bins are listed in
// `missed` (pair, first unit, units: a unit = kRescueUnit elements of the pair) with their exact bracket in `bm_rows`; this
// kernel re-reads those pairs ALONE and gathers the bracket's bins into list 1 — many workgroups per pair (a single
// workgroup pulls ~20 GB/s: the compaction route, whose workgroups own a fixed share of the batch, took 470 us for 70
// such pairs).  A persistent grid over the units; nothing to do (the usual case): every workgroup returns at once.
__global__ __launch_bounds__(kBlock) void k_octav_rescue_gather(const uint32_t* __restrict__ missed,
                                                                 const dpl_octav_state* __restrict__ ctl,
                                                                 dpl_octav_state* __restrict__ st,
                                                                 const dpl_span* __restrict__ pair_spans,
                                                                 const float* const* __restrict__ segs,
                                                                 const uint32_t* __restrict__ bm_rows,
                                                                 const uint64_t* __restrict__ pair_base,
                                                                 float* __restrict__ list1) {
    extern __shared__ __attribute__((aligned(16))) uint32_t queues[];
    __shared__ uint32_t bm[kKeyWords];
    __shared__ uint32_t found[3];
    const uint32_t n_missed = ctl->len[0], n_units = ctl->len[1];
    if (n_missed == 0u) return;
    const int w = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    uint32_t* q = queues + (size_t)w * kWave * kQueueStride + lane;
    uint32_t held = 0xFFFFFFFFu;   // the pair whose bitmap sits in bm
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < n_missed; e += kBlock) {   // which entry holds unit u (entries are in arrival order)
            const uint32_t u0 = missed[3 * e + 1], nu = missed[3 * e + 2];
            if (u - u0 < nu) {
                found[0] = missed[3 * e];
                found[1] = u - u0;
            }
        }
        __syncthreads();
        const uint32_t pair = found[0], c = found[1];
        dpl_octav_state* me = st + pair;
        if (pair != held) {
            for (int i = threadIdx.x; i < kKeyWords; i += kBlock) {
                const int j = i - kKeyWord0;
                bm[i] = (j >= 0 && j < kLogWords) ? bm_rows[(uint64_t)pair * kLogWords + j] : 0u;
            }
            held = pair;
            __syncthreads();
        }
        const dpl_span sp = pair_spans[pair];
        const uint64_t off = (uint64_t)c * kRescueUnit;
        const uint32_t cnt = (uint32_t)min((uint64_t)kRescueUnit, sp.count - off);
        gather_span(segs[sp.seg] + sp.offset + off, cnt, bm, q, me, reinterpret_cast<uint32_t*>(list1 + pair_base[pair]),
                    (uint32_t)(pair_base[pair + 1] - pair_base[pair]));
    }
}

// Per pair: the reference's exact iteration from the exact bin totals + the gathered elements.
// For the iterate s in bin jb:  count(|x| > s) = (exact total of the bins above jb) + (gathered values v of bin jb
// with v > s), and the same for the sums.  The gathered values are first bucketed by bin (one pass over list 0 into
// list 1: the bin's position comes from the exact counts, the rank inside the bin from an LDS counter), so that an
// iteration touches only the values of ITS bin - usually a few hundred, held in registers while the iterate stays
// in the bin - instead of walking the whole gathered list.  The totals above a bin are the suffix rows the bracket
// kernel left in the histogram buffers.  An iterate that lands in an unmarked bin sends the pair to the
// compaction route.
#ifndef DPL_EXACT_BLOCK
#define DPL_EXACT_BLOCK 128   // threads per pair
#endif
#ifndef DPL_EXACT_WAVES
#define DPL_EXACT_WAVES 4
#endif
#ifndef DPL_EXACT_REGS
#define DPL_EXACT_REGS 16
#endif
constexpr int kExactBlock = DPL_EXACT_BLOCK;
constexpr int kExactRegs = DPL_EXACT_REGS;  // values of the current bin held per lane (128 * 16 = 2 K)

__global__ __launch_bounds__(kExactBlock, DPL_EXACT_WAVES) void k_octav_exact(dpl_octav_state* __restrict__ st,
                                                              dpl_octav_state* __restrict__ ctl,
                                                              const uint32_t* __restrict__ pair_order,
                                                              const uint32_t* __restrict__ lh_cnt,
                                                              const unsigned long long* __restrict__ lh_sum,
                                                              const uint32_t* __restrict__ bitmap,
                                                              const uint64_t* __restrict__ pair_base,
                                                              const float* __restrict__ list0, float* __restrict__ list1,
                                                              int max_iters, int fail_every) {
    constexpr int kWaves = kExactBlock / kWave;
    constexpr int kPer = kLogNB / kExactBlock;     // consecutive bins owned by a thread in the offset scan
    static_assert(kLogNB % kExactBlock == 0, "bins must split evenly over the workgroup");
    __shared__ uint32_t boff[kLogNB];              // start of the bin's values inside the pair's list-1 region
    __shared__ uint32_t cursor[kLogNB];            // values placed so far (= the bin's count after the bucket pass)
    __shared__ double scr_s[kWaves];
    __shared__ uint32_t scr_n[kWaves];
    __shared__ uint32_t bm[kLogWords];
    __shared__ OctavStep s_step;
    __shared__ int s_jb;
    __shared__ uint32_t s_bad;
    const uint32_t pair = pair_order ? pair_order[blockIdx.x] : blockIdx.x;
    dpl_octav_state* me = st + pair;
    if (me->done || me->mode != 2u) return;  // uniform per workgroup
    const int w = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t* nge = lh_cnt + (uint64_t)pair * kLogNB;            // N_ge[j]
    const unsigned long long* sge = lh_sum + (uint64_t)pair * kLogNB;  // S_ge[j] (fp64 bits)
    if (threadIdx.x < kLogWords) bm[threadIdx.x] = bitmap[(uint64_t)pair * kBitmapRow + threadIdx.x];
    const unsigned long long n_elems = me->n_elems;
    const float unsigned_div = me->unsigned_div;
    const uint64_t base_off = pair_base[pair];
    float s = me->s;
    uint32_t iters = me->iters;
    const uint32_t n = me->len[0];
    __syncthreads();
    // ---- where each marked bin's values go: exclusive scan of the marked bins' exact counts, in bin order
    {
        const int b0 = (int)threadIdx.x * kPer;
        uint32_t cnt[kPer], local = 0u;
        uint32_t above = nge[b0];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int bq = b0 + q;
            const uint32_t next = (bq + 1 < kLogNB) ? nge[bq + 1] : 0u;
            const bool marked = (bm[bq >> 5] >> (bq & 31)) & 1u;
            cnt[q] = marked ? above - next : 0u;      // N_ge[b] - N_ge[b+1]
            above = next;
            local += cnt[q];
        }
        uint32_t inc = local;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, kWave);
            if (lane >= (uint32_t)o) inc += t;
        }
        if (lane == kWave - 1) scr_n[w] = inc;
        __syncthreads();
        uint32_t run = inc - local;
        for (int q = 0; q < w; ++q) run += scr_n[q];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            boff[b0 + q] = run;
            cursor[b0 + q] = 0u;
            run += cnt[q];
        }
    }
    __syncthreads();
    // ---- bucket pass: list 0 (as gathered) -> list 1 (grouped by bin); 16 values per thread in flight
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(list0 + base_off);   // 16-byte aligned region
        uint32_t* dst = reinterpret_cast<uint32_t*>(list1 + base_off);
        using u4 = __attribute__((ext_vector_type(4))) uint32_t;
        const u4* src4 = reinterpret_cast<const u4*>(src);
        const uint32_t n4 = n >> 2;
        auto place = [&](uint32_t u) {
            const int b = log_bin(__uint_as_float(u));
            const uint32_t r = atomicAdd(&cursor[b], 1u);
            dst[boff[b] + r] = u;
        };
        for (uint32_t i0 = threadIdx.x; i0 < n4; i0 += kExactBlock * 4) {
            u4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i = i0 + q * kExactBlock;
                v[q] = i < n4 ? __builtin_nontemporal_load(src4 + i) : u4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q * kExactBlock < n4) {
                    place(v[q].x);
                    place(v[q].y);
                    place(v[q].z);
                    place(v[q].w);
                }
            }
        }
        for (uint32_t i = (n4 << 2) + threadIdx.x; i < n; i += kExactBlock) place(src[i]);
    }
    // exact totals of the bins above the current one (thread 0): N_ge[j+1], S_ge[j+1]; nothing above the top bin
    int jb = log_bin(s);
    double s_above = 0.0;
    unsigned long long n_above = 0ull;
    auto load_above = [&](int j) {
        n_above = (j + 1 < kLogNB) ? (unsigned long long)nge[j + 1] : 0ull;
        s_above = (j + 1 < kLogNB) ? __longlong_as_double((long long)sge[j + 1]) : 0.0;
    };
    if (threadIdx.x == 0) {
        load_above(jb);
        s_jb = jb;
        s_bad = (jb <= 0 || jb >= kLogNB - 1 || !((bm[jb >> 5] >> (jb & 31)) & 1u)) ? 1u : 0u;
        if (fail_every > 0 && pair % (uint32_t)fail_every == 0u) s_bad = 1u;   // test hook: exercise the restart path
    }
    __syncthreads();   // list 1 and the counters are complete
    uint32_t done = 0u, bad = s_bad;
    int held = -1;             // the bin whose values are in r[]
    float r[kExactRegs];
    const float* grouped = list1 + base_off;
    while (!done && !bad) {
        const uint32_t nb = cursor[jb];
        const float* src = grouped + boff[jb];
        uint32_t c = 0;
        float p0 = 0.0f;
        double pd = 0.0;
        if (nb <= (uint32_t)(kExactBlock * kExactRegs)) {
            if (held != jb) {
#pragma unroll
                for (int j = 0; j < kExactRegs; ++j) {
                    const uint32_t idx = j * kExactBlock + threadIdx.x;
                    r[j] = idx < nb ? src[idx] : 0.0f;
                }
                held = jb;
            }
#pragma unroll
            for (int j = 0; j < kExactRegs; ++j) {
                const bool gq = r[j] > s;
                c += gq;
                p0 += gq ? r[j] : 0.0f;
            }
            pd = (double)p0;
        } else {  // a very full bin: walk its values from memory (L2) each time
            for (uint32_t i0 = 0; i0 < nb; i0 += kExactBlock * 16) {
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t idx = i0 + j * kExactBlock + threadIdx.x;
                    const float v = idx < nb ? src[idx] : 0.0f;
                    const bool gq = v > s;
                    c += gq;
                    part += gq ? v : 0.0f;
                }
                pd += (double)part;
            }
        }
        c = wave_sum(c);
        pd = wave_sum(pd);
        if (lane == 0) {
            scr_n[w] = c;
            scr_s[w] = pd;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tg = n_above;
            double ts = s_above;
            for (int j = 0; j < kWaves; ++j) {
                tg += scr_n[j];
                ts += scr_s[j];
            }
            const OctavStep q = octav_step(ts, tg, n_elems - tg, unsigned_div, s, iters, max_iters);
            s_step = q;
            if (!q.done) {
                const int jn = log_bin(q.s);
                if (jn <= 0 || jn >= kLogNB - 1 || !((bm[jn >> 5] >> (jn & 31)) & 1u)) {
                    s_bad = 1u;      // the bracket did not foresee this bin
                } else if (jn != s_jb) {
                    load_above(jn);
                    s_jb = jn;
                }
            }
        }
        __syncthreads();
        const OctavStep q = s_step;
        s = q.s;
        iters = q.iters;
        done = q.done;
        bad = done ? 0u : s_bad;
        jb = s_jb;
    }
    if (threadIdx.x == 0) {
        if (bad) {  // restart this pair on the compaction route from s_0 (state as k_octav_update<true> leaves it;
                    // s_0 is still in me->s: this kernel only writes it back on success)
            me->mode = 1u;
            me->iters = 0u;
            me->len[0] = 0u;
            me->len[1] = 0u;
            me->cur = 2u;
            atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
        } else {
            me->s = s;
            me->iters = iters;
            me->done = 1u;
        }
    }
}
}  // namespace

int g_exact_fail_every = 0;   // dpl_test_hook_exact_fail_every
int g_rescue_fail_every = 0;  // dpl_test_hook_rescue_fail_every

#ifndef DPL_RESCUE_GRID
#define DPL_RESCUE_GRID 512
#endif
// (octav_tail_host.hip) gather pass of the rescue: see k_octav_rescue_gather
int dpl_octav_rescue_gather_launch(const uint32_t* d_missed, dpl_octav_state* d_states, int64_t n_pairs, const dpl_span* d_pair_spans,
                                   const float* const* d_seg_ptrs, const uint32_t* d_bm_rows, const uint64_t* d_pair_base,
                                   float* d_list1, hipStream_t st) {
    hipLaunchKernelGGL(k_octav_rescue_gather, dim3(DPL_RESCUE_GRID), dim3(kBlock), (size_t)kBlock * kQueueStride * sizeof(uint32_t), st,
                       d_missed, d_states + n_pairs, d_states, d_pair_spans, d_seg_ptrs, d_bm_rows, d_pair_base, d_list1);
    DPL_LAUNCH_CHECK("k_octav_rescue_gather");
    return 0;
}

// The compaction route on its own, for the pairs a histogram form marked mode 1 (shared with octav_resident.hip).
int dpl_octav_fallback_route(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                             const float* const* d_seg_ptrs, dpl_octav_state* d_states, int64_t n_pairs,
                             const dpl_span* d_pair_spans, const uint64_t* d_pair_base, const uint32_t* d_pair_order,
                             float* d_list0, float* d_list1, int dynamic_sym, int max_iters, hipStream_t st) {
    const dim3 ug(grid_for(n_pairs, 256)), ub(256), pg((unsigned)n_blocks), pb(kBlock), pairs((unsigned)n_pairs);
    dpl_octav_state* ctl = d_states + n_pairs;
    const size_t stage_bytes = (size_t)(kBlock / kWave) * kStageCap * sizeof(float);
    hipLaunchKernelGGL(k_octav_compact_full, pg, pb, stage_bytes, st, d_items, d_block_begin, d_seg_ptrs, d_states,
                       ctl, d_pair_base, d_list0);
    hipLaunchKernelGGL(k_octav_update<false>, ug, ub, 0, st, d_states, n_pairs, dynamic_sym, max_iters, ctl);
    hipLaunchKernelGGL(k_octav_iterate_lists, pairs, dim3(kIterBlock),
                       (size_t)(kIterBlock / kWave) * kIterStageCap * sizeof(float), st, d_states, ctl, d_pair_order,
                       d_pair_base, d_list0, d_list1, max_iters);
    hipLaunchKernelGGL(k_octav_iterate_full, pairs, pb, 0, st, d_states, ctl, d_pair_spans, d_seg_ptrs, max_iters);
    DPL_LAUNCH_CHECK("k_octav_fallback_route");
    return 0;
}

extern "C" {

int dpl_test_hook_exact_fail_every(int every) {
    const int old = g_exact_fail_every;
    g_exact_fail_every = every;
    return old;
}

int dpl_test_hook_rescue_fail_every(int every) {
    const int old = g_rescue_fail_every;
    g_rescue_fail_every = every;
    return old;
}

int dpl_octav_init(dpl_octav_state* d_states, int64_t n_pairs, int list_mode, dpl_stream_t s) {
    if (n_pairs <= 0) return 0;
    if (list_mode < 0 || list_mode > 2) return fail_msg("dpl_octav_init: mode must be 0 (full), 1 (compaction) or 2 (bracket)");
    hipLaunchKernelGGL(k_octav_init, dim3(grid_for(n_pairs + 1, 256)), dim3(256), 0, (hipStream_t)s, d_states,
                       n_pairs, (uint32_t)list_mode);
    DPL_LAUNCH_CHECK("k_octav_init");
    return 0;
}

int dpl_octav_run(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                  const float* const* d_seg_ptrs, dpl_octav_state* d_states, int64_t n_pairs, int dynamic_sym,
                  int max_iters, dpl_stream_t s) {
    if (n_items <= 0 || n_pairs <= 0) return 0;
    if (int e = check_blocks("dpl_octav_run", n_items, d_block_begin, n_blocks)) return e;
    hipStream_t st = (hipStream_t)s;
    const dim3 ug(grid_for(n_pairs, 256)), ub(256), pg((unsigned)n_blocks), pb(kBlock);
    dpl_octav_state* ctl = d_states + n_pairs;
    hipLaunchKernelGGL(k_octav_pass<true>, pg, pb, 0, st, d_items, d_block_begin, d_seg_ptrs, d_states, ctl);
    hipLaunchKernelGGL(k_octav_update<true>, ug, ub, 0, st, d_states, n_pairs, dynamic_sym, max_iters, ctl);
    for (int k = 0; k < max_iters; ++k) {
        hipLaunchKernelGGL(k_octav_pass<false>, pg, pb, 0, st, d_items, d_block_begin, d_seg_ptrs, d_states, ctl);
        hipLaunchKernelGGL(k_octav_update<false>, ug, ub, 0, st, d_states, n_pairs, dynamic_sym, max_iters, ctl);
    }
    DPL_LAUNCH_CHECK("k_octav");
    return 0;
}

int dpl_octav_run_compact(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, dpl_octav_state* d_states,
                          int64_t n_pairs, const dpl_span* d_pair_spans, const uint64_t* d_pair_base,
                          const uint32_t* d_pair_order, float* d_list0, float* d_list1, int dynamic_sym,
                          int max_iters, dpl_stream_t s) {
    if (n_items <= 0 || n_pairs <= 0) return 0;
    if (int e = check_blocks("dpl_octav_run_compact", n_items, d_block_begin, n_blocks)) return e;
    hipStream_t st = (hipStream_t)s;
    const dim3 ug(grid_for(n_pairs, 256)), ub(256), pg((unsigned)n_blocks), pb(kBlock);
    dpl_octav_state* ctl = d_states + n_pairs;
    // 1. statistics + s_0             2. evaluate at s_0 over the full data, keep the values above s_0
    hipLaunchKernelGGL(k_octav_pass<true>, pg, pb, 0, st, d_items, d_block_begin, d_seg_ptrs, d_states, ctl);
    hipLaunchKernelGGL(k_octav_update<true>, ug, ub, 0, st, d_states, n_pairs, dynamic_sym, max_iters, ctl);
    if (max_iters > 0) {
        hipLaunchKernelGGL(k_octav_compact_full, pg, pb, (size_t)(kBlock / kWave) * kStageCap * sizeof(float), st,
                           d_items, d_block_begin, d_seg_ptrs, d_states, ctl, d_pair_base, d_list0);
        hipLaunchKernelGGL(k_octav_update<false>, ug, ub, 0, st, d_states, n_pairs, dynamic_sym, max_iters, ctl);
        // 3. every remaining iteration of every pair inside one launch    4. degenerate pairs on the full data
        hipLaunchKernelGGL(k_octav_iterate_lists, dim3((unsigned)n_pairs), dim3(kIterBlock),
                           (size_t)(kIterBlock / kWave) * kIterStageCap * sizeof(float), st, d_states, ctl, d_pair_order,
                           d_pair_base, d_list0, d_list1, max_iters);
        hipLaunchKernelGGL(k_octav_iterate_full, dim3((unsigned)n_pairs), pb, 0, st, d_states, ctl, d_pair_spans,
                           d_seg_ptrs, max_iters);
    }
    DPL_LAUNCH_CHECK("k_octav_compact");
    return 0;
}

int dpl_octav_run_bracket(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, dpl_octav_state* d_states,
                          int64_t n_pairs, const dpl_span* d_pair_spans, const uint64_t* d_pair_base,
                          const uint32_t* d_pair_order, float* d_list0, float* d_list1, uint32_t* d_lh_cnt,
                          uint64_t* d_lh_sum, uint32_t* d_bitmap, int dynamic_sym, int max_iters, dpl_stream_t s) {
    if (n_items <= 0 || n_pairs <= 0) return 0;
    if (int e = check_blocks("dpl_octav_run_bracket", n_items, d_block_begin, n_blocks)) return e;
    hipStream_t st = (hipStream_t)s;
    const dim3 pg((unsigned)n_blocks), pb(kBlock), pairs((unsigned)n_pairs);
    dpl_octav_state* ctl = d_states + n_pairs;
    hipError_t e1 = hipMemsetAsync(d_lh_cnt, 0, (size_t)n_pairs * kLogNB * sizeof(uint32_t), st);
    hipError_t e2 = hipMemsetAsync(d_lh_sum, 0, (size_t)n_pairs * kLogNB * sizeof(uint64_t), st);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail("hipMemsetAsync", e1 != hipSuccess ? e1 : e2);
    // 1. statistics + log-scale histogram   2. s_0 and the bracket walk   3. gather the marked bins   4. exact walk
    hipLaunchKernelGGL(k_octav_loghist, pg, pb, (size_t)kLogNB * 8, st, d_items, d_block_begin, d_seg_ptrs, d_states,
                       d_lh_cnt, reinterpret_cast<unsigned long long*>(d_lh_sum));
    hipLaunchKernelGGL(k_octav_bracket, pairs, pb, 0, st, d_states, ctl, d_lh_cnt,
                       reinterpret_cast<unsigned long long*>(d_lh_sum), d_bitmap, dynamic_sym, max_iters);
    if (max_iters > 0) {
        hipLaunchKernelGGL(k_octav_gather, pg, pb, (size_t)kBlock * kQueueStride * sizeof(uint32_t), st, d_items,
                           d_block_begin, d_seg_ptrs, d_states, d_bitmap, d_pair_base, d_list0);
        hipLaunchKernelGGL(k_octav_exact, pairs, dim3(kExactBlock), 0, st, d_states, ctl, d_pair_order,
                           d_lh_cnt, reinterpret_cast<const unsigned long long*>(d_lh_sum), d_bitmap, d_pair_base,
                           d_list0, d_list1, max_iters, g_exact_fail_every);
        // 5. pairs the bracket could not serve (flat / degenerate distributions, values >= 2^14): compaction route
        if (int e = dpl_octav_fallback_route(d_items, n_items, d_block_begin, n_blocks, d_seg_ptrs, d_states, n_pairs,
                                             d_pair_spans, d_pair_base, d_pair_order, d_list0, d_list1, dynamic_sym, max_iters, st))
            return e;
    }
    DPL_LAUNCH_CHECK("k_octav_bracket");
    return 0;
}

int dpl_octav_finalize(const dpl_octav_state* d_states, int64_t n_pairs, float* d_out, dpl_stream_t s) {
    if (n_pairs <= 0) return 0;
    hipLaunchKernelGGL(k_octav_finalize, dim3(grid_for(n_pairs, 256)), dim3(256), 0, (hipStream_t)s, d_states,
                       n_pairs, d_out);
    DPL_LAUNCH_CHECK("k_octav_finalize");
    return 0;
}
}  // extern "C"
