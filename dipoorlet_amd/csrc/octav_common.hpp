// Shared by the OCTAV translation units (octav_kernels.hip: full / compaction / two-read bracket forms;
// octav_resident.hip: the single-read register-resident form): the fixed-point step, the log-scale histogram
// geometry and the bracket walk over its bin edges.
#pragma once
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

// One fixed-point step (forward_net.py:326-330): s' = fl32(sum) / fl32(c/unsigned * cnt_le + cnt_gt) — the python-float
// denominator is cast to float32 for the divide (NEP 50); |s' - s| < 1e-6 stops KEEPING the previous s.
struct OctavStep {
    float s;
    uint32_t iters, done, decreased;
};
__device__ __forceinline__ OctavStep octav_step(double sum, unsigned long long cnt_gt, unsigned long long cnt_le,
                                                float unsigned_div, float s, uint32_t iters, int max_iters) {
    const double c = 1.0 / 65536.0 / 3.0 / (double)unsigned_div;
    const double denom = c * (double)(long long)cnt_le + (double)(long long)cnt_gt;
    const float s1 = __fdiv_rn((float)sum, (float)denom);
    OctavStep r{s, iters, 0u, 0u};
    if (fabsf(__fsub_rn(s1, s)) < 1e-6f) {
        r.done = 1u;
    } else {
        r.decreased = !(s1 >= s) ? 1u : 0u;
        r.s = s1;
        r.iters = iters + 1u;
        if ((int)r.iters >= max_iters || s1 != s1) r.done = 1u;
    }
    return r;
}


#ifndef DPL_MARGIN0
#define DPL_MARGIN0 0
#endif
#ifndef DPL_MARGIN
#define DPL_MARGIN 0
#endif
constexpr int kLogNB = 2048;
constexpr int kLogShift = 17;                               // 23 - 6: six mantissa bits per bin
constexpr uint32_t kLogKey0 = (uint32_t)(127 - 18) << 6;    // key of 2^-18
constexpr int kLogWords = kLogNB / 32;
constexpr int kBitmapRow = kLogWords + 2;                   // + the gather range [lo, hi) as float bits
constexpr int kLogMaxMarked = 256;
constexpr uint32_t kSmallPair = 16384;                      // pairs this small are gathered whole
constexpr uint32_t kRescueUnit = 16384;                     // elements of a pair one workgroup of k_octav_rescue_gather re-reads

// Capacity (elements, a multiple of 32: whole 128-byte lines) of the LIST REGION of one slice of n elements in the one-read
// forms' list buffers.  The exact-tail form lists ~0.5 - 1.5 % of a pair (a wave's budget: kTailAllow0 + what it has seen >> 6,
// + 512 per raise) and the rescue gathers a bracket's bins (~2 %): a region holds n / 16 + 16384 values, never more than the
// slice itself.  What does not fit — saturating activations with a tenth of their values at the maximum, constant tensors —
// is not listed: the pair's walk is refused (its list length says so) and it finishes on the compaction route, whose
// full-size lists the caller provides only when a batch reports such pairs.  Round 4 gave every pair a region of its own
// size in every list: 4 x the batch's activations in scratch.  (n / 32 + 16384 was too tight for the rescue: the bracket of a
// cold 802 816-element pair holds 5 - 6 % of it, and 14 pairs of every cold ResNet-50 sweep ended on the compaction route.)
constexpr uint32_t kListCapShift = 4, kListCapConst = 16384;
constexpr uint32_t kListWhole = 20480;   // a pair this small lists its whole window (octav_tail_host.hip: kSmallCap): its region holds all of it
__host__ __device__ inline uint32_t list_cap_of(unsigned long long n) {
    const unsigned long long whole = (n + 31ull) & ~31ull, part = ((n >> kListCapShift) + kListCapConst + 31ull) & ~31ull;
    return (uint32_t)((whole < part || n <= kListWhole) ? whole : part);
}

__device__ __forceinline__ int log_bin(float a) {
    const int b = (int)(__float_as_uint(a) >> kLogShift) - (int)kLogKey0;
    return b < 0 ? 0 : (b > kLogNB - 1 ? kLogNB - 1 : b);
}
__device__ __forceinline__ double log_bin_scale(int b) {   // 2^(e - 150) for the exponent field e of bin b
    const int e = (int)(((uint32_t)b + kLogKey0) >> 6);
    return __longlong_as_double((long long)(e - 150 + 1023) << 52);
}

// One 64-bit LDS atomic per element: the bin word holds the count in bits 43..62 and the mantissa sum in bits
// 0..42 (a sub-span has < 2^20 elements, an explicit mantissa is < 2^23: neither field can overflow into the other).
constexpr int kPackShift = 43;   // count: bits 43..62 (a slice has < 2^20 elements), bit 63 stays free for the gather flag
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;


__device__ __forceinline__ float log_edge(int b) {  // lower edge of bin b (bin 0 starts at 0)
    return b <= 0 ? 0.0f : __uint_as_float(((uint32_t)b + kLogKey0) << kLogShift);
}


// s_0 and the bracket walk of one pair (one thread).  n_ge[j] / s_ge[j] = exact count / sum of the window values in
// bins >= j (entries 1 .. kLogNB-1 are read); sum_out / nz_out = the directly accumulated nonzero values outside the
// binned window.  Marks in bm[kLogWords] (zeroed by the caller) the bins the true iterates can fall into.
//   route 0: the pair is finished already (NaN iterate, or no iterations asked for)
//   route 2: bracket route — gather the marked bins [jmin, jmax], then walk the exact iteration
//   route 1: compaction route (values >= 2^14 or inf, a bracket that leaves the window or marks too many bins)
struct BracketResult {
    float s0, unsigned_div;
    uint32_t route;
    int jmin, jmax;
};
// The bracket walk proper: from s_0 (finite) over the bin edges, marking in bm[kLogWords] (zeroed by the caller) the bins
// the true iterates can fall into.  Returns route 2 (marks valid, bins [jmin, jmax]) or 1 (the bracket leaves the window
// or marks more than kLogMaxMarked bins: such a pair belongs on the compaction route).
__device__ __forceinline__ BracketResult bracket_marks(const uint32_t* n_ge, const double* s_ge, uint32_t* bm, float s0,
                                                       float ud, unsigned long long n) {
    BracketResult out;
    out.s0 = s0;
    out.unsigned_div = ud;
    out.jmin = kLogNB;
    out.jmax = -1;
    uint32_t r = 2u;
    const double c = 1.0 / 65536.0 / 3.0 / (double)ud;
    double lo = (double)s0, hi = (double)s0;
    int marked = 0;
    for (int itn = 0; itn < 20 && r == 2u; ++itn) {
        const int jl = log_bin((float)lo), jh = log_bin((float)hi);
        if (jl <= 1 || jh >= kLogNB - 2 || !(lo == lo) || !(hi == hi)) {
            r = 1u;
            break;
        }
        // Mark exactly the bins of the bracket, no margin.  Within a bin F(s) moves monotonically between its
        // values at the two edges unless the bin contains F itself (dropping a value v raises F iff v < F),
        // i.e. only at the fixed point, where the excursion beyond the edge values is second order; together
        // with the fp32 rounding of the true iterate that can put an iterate one bin outside the bracket
        // with a probability of order 1e-4 per pair.  The exact walk verifies every iterate and such a pair
        // simply finishes on the compaction route; a margin bin on either side (DPL_MARGIN=1) would more
        // than double the values gathered (2.2 % -> 4.9 % on ResNet-50 activations) to avoid that.
        const int ml = jl - (itn == 0 ? DPL_MARGIN0 : DPL_MARGIN), mh = jh + (itn == 0 ? DPL_MARGIN0 : DPL_MARGIN);
        for (int w0 = ml >> 5; w0 <= mh >> 5; ++w0) {   // one LDS read-modify-write per word
            const int lo_b = max(ml, w0 << 5) & 31, hi_b = min(mh, (w0 << 5) + 31) & 31;
            const uint32_t mask = (0xFFFFFFFFu >> (31 - hi_b)) & (0xFFFFFFFFu << lo_b);
            const uint32_t old = bm[w0];
            bm[w0] = old | mask;
            marked += __popc(mask & ~old);
        }
        out.jmin = ml < out.jmin ? ml : out.jmin;
        out.jmax = mh > out.jmax ? mh : out.jmax;
        if (marked > kLogMaxMarked) {
            r = 1u;
            break;
        }
        double nlo = INFINITY, nhi = -INFINITY;
        for (int j = jl; j <= jh + 1; ++j) {  // F with everything in bins >= j counted as "above"
            // An edge above every value of the pair is no bound: no iterate reaches it (F is a mean of the pair's values), and F
            // there — 0 — would drag the bracket down to the bottom of the window.  Seen on erf outputs, whose values pile up
            // just below 1: the iteration ends inside that top bin.  (An iterate that does fall below the marked bins — few
            // values above it — is caught by the exact walk like any other.)
            if (n_ge[j] == 0u) continue;
            const double ng = (double)n_ge[j];
            const double f = s_ge[j] / (c * ((double)(long long)n - ng) + ng);
            nlo = fmin(nlo, f);
            nhi = fmax(nhi, f);
        }
        if (!(nlo <= nhi)) break;           // nothing above the bracket's bins at all
        if (nlo == lo && nhi == hi) break;  // the bracket stopped moving
        lo = nlo;
        hi = nhi;
    }
    out.route = r;
    return out;
}

__device__ __forceinline__ BracketResult bracket_walk(const uint32_t* n_ge, const double* s_ge, uint32_t* bm, float mn,
                                                      float mx, bool nan_seen, double sum_out,
                                                      unsigned long long nz_out, unsigned long long n, int dynamic_sym,
                                                      int max_iters) {
    // forward_net.py:319 — np.abs(data_min - 0) < 1e-6 (float32 compare) and 'dynamic_sym' in qi_params
    const float ud = (dynamic_sym && fabsf(mn) < 1e-6f && !nan_seen) ? 4.0f : 1.0f;
    // forward_net.py:324 — sum(|x|) / count(|x| > 0): exact window totals + the out-of-window part
    const float s0 = nan_seen ? __uint_as_float(0x7FC00000u)
                              : __fdiv_rn((float)(sum_out + s_ge[1]), (float)(long long)(nz_out + n_ge[1]));
    const float max_abs = fmaxf(fabsf(mn), fabsf(mx));
    BracketResult out;
    out.s0 = s0;
    out.unsigned_div = ud;
    out.jmin = kLogNB;
    out.jmax = -1;
    out.route = 2u;
    if (s0 != s0 || max_iters <= 0) {
        out.route = 0u;  // NaN is a fixed point of the iteration
    } else if (!(max_abs < log_edge(kLogNB))) {
        out.route = 1u;  // values at or above 2^14 (or inf): outside the exactly-summed window
    } else if (n <= (unsigned long long)kSmallPair) {
        for (int q = 0; q < kLogWords; ++q) bm[q] = 0xFFFFFFFFu;  // gather the whole (small) pair's window
        out.jmin = 0;
        out.jmax = kLogNB - 1;
    } else {
        out = bracket_marks(n_ge, s_ge, bm, s0, ud, n);
    }
    return out;
}

__device__ __forceinline__ void load_tile(const float* __restrict__ p_generic, uint32_t base, uint32_t n, bool aligned,
                                          f4 (&v)[4]) {
    gptr_f32 p = (gptr_f32)p_generic;
    const uint32_t lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t idx = base + u * 256 + lane * 4;
        if (aligned && idx + 3 < n) {
            v[u] = __builtin_nontemporal_load((gptr_f4)(p + idx));
        } else {  // zeros never survive (s >= 0)
            v[u].x = idx + 0 < n ? p[idx + 0] : 0.0f;
            v[u].y = idx + 1 < n ? p[idx + 1] : 0.0f;
            v[u].z = idx + 2 < n ? p[idx + 2] : 0.0f;
            v[u].w = idx + 3 < n ? p[idx + 3] : 0.0f;
        }
    }
}

// Tile walker for the compaction-style kernels: wave w of the workgroup takes the 1024-element tiles
// w, w + waves, ... of p[0..n).  Full tiles of an aligned span go through a branch-free, software-pipelined
// loop (two register sets; the next tile's four 16-byte loads are in flight while the current tile is
// consumed); the ragged end (< one workgroup tile) or an unaligned span uses the bounds-checked loader, which
// pads with zeros.  eat(v, tile_base, full): `full` tells the consumer that no element is padding.
template <int kThreads, class Eat>
__device__ __forceinline__ void for_each_tile(const float* __restrict__ p, uint32_t n, Eat&& eat) {
    const uint32_t w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    constexpr uint32_t kStep = kThreads * 16;
    const bool aligned = (((uintptr_t)p) & 15u) == 0;
    const uint32_t n_main = aligned ? (n / kStep) * kStep : 0u;
    gptr_f4 pv = (gptr_f4)p;
#define DPL_TLOAD(buf, t0)                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) buf[u] = __builtin_nontemporal_load(pv + ((t0) >> 2) + u * 64 + lane)
    uint32_t tile = w * 1024;
    if (tile < n_main) {
        f4 A[4], B[4];
        DPL_TLOAD(A, tile);
        for (;;) {
            uint32_t nxt = tile + kStep;
            if (nxt >= n_main) {
                eat(A, tile, true);
                break;
            }
            DPL_TLOAD(B, nxt);
            eat(A, tile, true);
            tile = nxt;
            nxt = tile + kStep;
            if (nxt >= n_main) {
                eat(B, tile, true);
                break;
            }
            DPL_TLOAD(A, nxt);
            eat(B, tile, true);
            tile = nxt;
        }
    }
#undef DPL_TLOAD
    for (uint32_t t2 = n_main + w * 1024; t2 < n; t2 += kStep) {
        f4 v[4];
        load_tile(p, t2, n, aligned, v);
        eat(v, t2, false);
    }
}


}  // namespace
