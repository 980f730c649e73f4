// OCTAV ('-A mse', forward_net.py:284-342) in ONE read of the activations: the host side of the exact-tail form and what it shares
// with its rescue.
//
// Why not two reads: measured on MI355X (scripts/mall_probe.hip, profiles/r02/mall_probe.txt) a re-read of recently
// streamed data costs the same whether HBM or the 256 MiB Infinity Cache serves it (6.1-6.9 TB/s either way, one
// shared fabric), so the two-read bracket form of octav_kernels.hip cannot pass ~40 % of the roofline.  And a form that
// keeps a pair on chip until a leader has walked its bracket (tried first, round 2) spends its time waiting.
//
// What is here:
//   octav_tail.hpp (included below)   k_octav_tail / k_octav_tail_merge / k_octav_tail_init: one workgroup per slice streams it
//                        (min / max, exact log-scale histogram in LDS, the values at or above a threshold bin listed) and walks
//                        the pair — early iterates as lower bounds from the histogram, late ones exactly from the list;
//   walk_rescued + k_octav_walk_rescue   the RESCUE of a pair whose walk was refused: the reference's whole iterate sequence on
//                        (exact totals of the bins above) + (the values of the pair's exact bracket, re-read by
//                        k_octav_rescue_gather in octav_kernels.hip), every iterate verified;
//   dpl_octav_oneread_* / dpl_octav_plan_*   the C ABI: one job struct per batch; a HOST plan that sizes, lays out and binds it.
// Rounds 2 - 3 listed the bins ALL iterates were predicted to visit (k_octav_oneread, k_octav_probe, k_octav_sort,
// k_octav_walk[_sorted]: DESIGN 3c, 3d); the exact-tail form superseded them in round 4 and round 5 removed them.
// No workgroup ever waits for another; what crosses kernels crosses launches.
#include <type_traits>
#include "common.hpp"
#include "octav_common.hpp"

#pragma clang fp contract(off)

namespace {

// keeps the scheduler from interleaving the unrolled per-vector bodies (their temporaries would not fit beside the slice)
#define DPL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
#ifndef DPL_RES_VEC
#define DPL_RES_VEC 16
#endif
#ifndef DPL_WALK_OCC
#define DPL_WALK_OCC 4
#endif
constexpr int kVec = DPL_RES_VEC;                               // 16-byte vectors per thread the rescue walk keeps a list in
constexpr int kOver = 4;                                         // rows of a list beyond the resident ones streamed per step of an iteration
constexpr uint32_t kSmallCap = 20480;                           // pairs this small list their whole window (every step of their walk is exact)
static_assert(kSmallCap == kListWhole, "a small pair's list region must hold the whole pair (list_cap_of)");
#ifndef DPL_SLICE_CAP
#define DPL_SLICE_CAP 1044480
#endif
constexpr uint32_t kCap = DPL_SLICE_CAP;                        // elements of a slice (streamed tile by tile)
static_assert(kCap < (1u << 20) && kCap % 4096u == 0u, "a slice's bin counts must fit the packed field");
constexpr uint32_t kMaxCluster = 64;                            // slices of one pair at most
constexpr int64_t kPlanEpoch = 8;   // batches per threshold-history epoch (dpl_octav_plan_bind)
#ifndef DPL_RESCUE_GRID
#define DPL_RESCUE_GRID 512
#endif
constexpr unsigned kRescueGrid = DPL_RESCUE_GRID;   // workgroups of the rescue's persistent kernels
constexpr int kPredRow = 2 * kLogWords;             // u32 words of a tensor's row in d_pred (word 0: the threshold snapshot)
constexpr int kRescRow = kLogNB + kLogNB / 2;   // u64 words of a rescued pair's row: 2048 suffix sums (fp64) + 2048 suffix counts (u32)

// LDS of the streaming kernel: [A: packed histogram 16 KiB | one dummy word per lane][B: the waves' survivor queues, octav_tail.hpp]
constexpr int kLdsA = kLogNB * 8 + kWave * 8;                   // + the lanes' dummy words

#ifdef DPL_RES_PROF
// phase cycle counters of a tuning build (scripts/res_prof.py): [workgroup][8] u64, accumulated by thread 0
__device__ unsigned long long g_res_prof[4096 * 8];
#define DPL_PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define DPL_PROF_ADD(slot, a, b) do { if (threadIdx.x == 0) g_res_prof[(blockIdx.x & 4095u) * 8 + (slot)] += (b) - (a); } while (0)
__device__ __forceinline__ void g_prof_iters_add(uint32_t b, uint32_t it) { g_res_prof[(b & 4095u) * 8 + 7] += it; }
#define DPL_PROF_L(len) g_res_prof[(blockIdx.x & 4095u) * 8 + 6] += (len)
#else
#define DPL_PROF_L(len) do {} while (0)
__device__ __forceinline__ void g_prof_iters_add(uint32_t, uint32_t) {}
#define DPL_PROF_T(var) do {} while (0)
#define DPL_PROF_ADD(slot, a, b) do {} while (0)
#endif

struct Shared {
    double red_d[kWaves];
    unsigned long long red_q[kWaves];
    uint32_t red_a[kWaves], red_b[kWaves];
    unsigned long long part_m[2][kWaves];   // walk: the waves' partial (count, mantissa sum), two alternating slots
    uint32_t part_c[2][kWaves];
    float red_mn[kWaves], red_mx[kWaves];
    double low_sum;               // streaming kernel: non-zero values outside the window: their sum, count, a NaN among them
    uint32_t low_cnt, low_nan;
    uint32_t bm[kLogWords];       // rescue walk: the bins of the pair's bracket (their values were gathered)
    uint32_t pub[kLogWords];      // exact-tail walk: the bracket of a refused pair (bracket_marks)
    uint32_t would_list;          // ... and the values its marked bins hold (must fit the pair's region of the rescue list)
    uint32_t cursor;              // streaming kernel: entries of the slice's list region handed out so far
    uint32_t list_cap, region_cap;   // exact-tail form: values this workgroup's list part / the pair's whole list region holds
    uint32_t tail_j;              // exact-tail form (octav_tail.hpp): the bin at and above which values are listed (only ever raised)
    uint32_t jwant;               // ... and the bin this pair asks the tensor's next batches to list from
    // ... wave 0 walks alone; what it hands to the others (and to the pair's state) at the joints of the walk
    float t_s, w_s0, w_ud;
    int w_jb;
    uint32_t w_evals, w_exact, w_path, w_bad, w_route, w_lkn, w_lkc;
    double w_lks;
    uint32_t seg_off[kMaxCluster], seg_len[kMaxCluster];   // merge: the slices' list segments; rescue walk: [0] = the list's length
    OctavStep step;
    int jb;
    uint32_t bad, route;
    float s0, ud, w_s;
    double s_above;
    unsigned long long n_above, n_elems;
};

// wave64 inclusive prefix sums by DPP (Hillis-Steele inside each row of 16, then the two row broadcasts): VALU only — the
// ds_bpermute form (__shfl_up) is six dependent trips through the LDS pipeline per value, which inside the streaming kernel is
// full of the other workgroups' histogram atomics
__device__ __forceinline__ uint32_t scan_u32_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__device__ __forceinline__ double scan_f64_dpp(double v) {
#define DPL_SCAN_STEP(ctrl, rmask, bound)                                                                              \
    {                                                                                                                  \
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);                                      \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, ctrl, rmask, 0xF, bound);       \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), ctrl, rmask, 0xF, bound); \
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                   \
    }
    DPL_SCAN_STEP(0x111, 0xF, true)
    DPL_SCAN_STEP(0x112, 0xF, true)
    DPL_SCAN_STEP(0x114, 0xF, true)
    DPL_SCAN_STEP(0x118, 0xF, true)
    DPL_SCAN_STEP(0x142, 0xA, false)
    DPL_SCAN_STEP(0x143, 0xC, false)
#undef DPL_SCAN_STEP
    return v;
}

// Raw per-bin (count, scaled sum) in n_ge / s_ge -> suffix totals in place (N_ge[j], S_ge[j] = everything in bins >= j).
// Thread t owns the 8 bins below 2047 - 8 t; all 256 threads; the raw values were written by their owners.
__device__ __forceinline__ void suffix_in_place(uint32_t* n_ge, double* s_ge, Shared& sh) {
    constexpr int kPerT = kLogNB / kThreads;
    const int hi = kLogNB - 1 - (int)threadIdx.x * kPerT;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    uint32_t ln = 0;
    double ls = 0.0;
    for (int q = 0; q < kPerT; ++q) {
        ln += n_ge[hi - q];
        ls += s_ge[hi - q];
    }
    const double is = scan_f64_dpp(ls);
    const uint32_t in = scan_u32_dpp(ln);
    if (lane == kWave - 1) {
        sh.red_d[w] = is;
        sh.red_a[w] = in;
    }
    __syncthreads();
    double rs = is - ls;
    uint32_t rn = in - ln;
    for (int q = 0; q < w; ++q) {
        rs += sh.red_d[q];
        rn += sh.red_a[q];
    }
    for (int q = 0; q < kPerT; ++q) {
        const int b = hi - q;
        rn += n_ge[b];
        rs += s_ge[b];
        n_ge[b] = rn;
        s_ge[b] = rs;
    }
    __syncthreads();
}

__device__ __forceinline__ double bin_sum(unsigned long long mant_explicit, uint32_t count, int b) {
    return (double)(mant_explicit + ((unsigned long long)count << 23)) * log_bin_scale(b);   // full 24-bit mantissas
}

// (the dynamic LDS block is addressed through address-space-3 pointers: ds_ instructions with constant offsets)
typedef __attribute__((address_space(3))) unsigned long long* lptr_u64;
typedef __attribute__((address_space(3))) uint32_t* lptr_u32;
// wave64 sum by DPP (row-local butterflies, then the two row broadcasts): ~6 VALU instead of six dependent ds_bpermute round
// trips; the total arrives in lane 63 and is broadcast from there
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror: every lane holds its row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The RESCUE walk of one pair (one workgroup; phase 2 of rounds 3 - 4's walk_pair, which also served the forms that round 5
// removed): a pair whose exact-tail walk was refused restarts from s_0 and walks the reference's WHOLE iterate sequence
// (forward_net.py:325-330) — totals of the bins above the iterate's bin from the suffix totals the first walk saved (exact
// integers), the values of the iterate's own bin from the list k_octav_rescue_gather collected (integer mantissa sums) —
// verifying that every iterate lands in a bin of the pair's bracket (rescue_bm).  An iterate outside it, or a list longer than
// the pair's region of the rescue list (it was cut), hands the pair to the compaction route.
template <int kVecT>
__device__ __forceinline__ void walk_rescued(
    const uint32_t pair, double* s_ge, uint32_t* n_ge, Shared& sh, dpl_octav_state* __restrict__ st,
    dpl_octav_state* __restrict__ ctl, const uint64_t* __restrict__ pair_base, int max_iters, int fail_every,
    const uint32_t* __restrict__ rescue_bm, const float* __restrict__ list_rescue, const unsigned long long* __restrict__ resc) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    dpl_octav_state* me = st + pair;
    if (me->mode != 3u || me->done) return;
    if (me->n_elems == 0ull) return;   // an empty pair: nothing was streamed
    DPL_PROF_T(pt0);
    const float* lp = list_rescue + pair_base[pair];
    f4 v[kVecT];
    // the list: ONE segment at the start of the pair's region; 1024 values per ROW (one 16-byte vector per thread)
    auto load_rows = [&](auto& dst, auto count, uint32_t row0, uint32_t len) {
        constexpr int kN = decltype(count)::value;
        const uint32_t voff = tid << 4;
#pragma unroll
        for (int u = 0; u < kN; ++u) {
            const uint32_t e0 = (row0 + (uint32_t)u) << 10;
            // buffer loads: zero fill past the list's end (one descriptor per row: the range check leaves the SGPR offset out,
            // so the row offset goes into the base)
            const int nbytes = e0 < len ? (int)(min(len - e0, 1024u) << 2) : 0;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(lp + (e0 < len ? e0 : 0u)), 0, nbytes, 0x00020000);
            dst[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        }
    };
    {   // the suffix totals the first walk left for this pair (own bins per thread)
        constexpr int kPerT = kLogNB / kThreads;
        const int hi = kLogNB - 1 - (int)tid * kPerT;
        const double* rs = reinterpret_cast<const double*>(resc + (uint64_t)pair * kRescRow);
        const uint32_t* rn = reinterpret_cast<const uint32_t*>(resc + (uint64_t)pair * kRescRow + kLogNB);
#pragma unroll
        for (int qq = 0; qq < kPerT; ++qq) {
            n_ge[hi - qq] = rn[hi - qq];
            s_ge[hi - qq] = rs[hi - qq];
        }
    }
    if (tid < (uint32_t)kLogWords) sh.bm[tid] = rescue_bm[(uint64_t)pair * kLogWords + tid];   // the bins whose values were gathered
    if (tid == 0) {   // s_0 and the divisor are in the state since the first walk
        sh.s0 = me->s;
        sh.ud = me->unsigned_div;
        sh.n_elems = me->n_elems;
        sh.seg_len[0] = me->len[0];
        // (more gathered than the pair's region of the rescue list holds: the list is incomplete — the compaction route)
        sh.route = me->len[0] > (uint32_t)(pair_base[pair + 1] - pair_base[pair]) ? 1u : 2u;
    }
    __syncthreads();
    DPL_PROF_T(pt1);
    DPL_PROF_ADD(0, pt0, pt1);
    const uint32_t route = __builtin_amdgcn_readfirstlane(sh.route);
    uint32_t bad = route == 1u ? 1u : 0u;
    float s = sh.s0;
    uint32_t iters = 0u;
    if (route == 2u) {
        const float ud = sh.ud;
        const unsigned long long n_elems = sh.n_elems;
        const uint32_t L = __builtin_amdgcn_readfirstlane(sh.seg_len[0]);
        const uint32_t n_rows = (L + 1023u) >> 10;
        // the first kVec rows stay in registers for the whole walk; the rows beyond are streamed kOver at a time in every
        // iteration — requested before the resident rows are scanned, consumed after
        f4 ov[kOver];
        load_rows(v, std::integral_constant<int, kVecT>{}, 0u, L);
        // every wave takes the step itself from the four partial sums (one barrier and two LDS round trips per iteration); the
        // gathered-bin bitmap sits in registers (lane l: word l)
        const uint32_t bm_reg = sh.bm[lane];
        auto marked = [&](int j) {
            return j > 0 && j < kLogNB - 1 && (((uint32_t)__builtin_amdgcn_readlane((int)bm_reg, j >> 5) >> (j & 31)) & 1u);
        };
        int jb = log_bin(s);
        bad = marked(jb) ? 0u : 1u;
        if (fail_every > 0 && pair % (uint32_t)fail_every == 0u) bad = 1u;   // test hook: the compaction route
        unsigned long long n_above = 0ull;
        double s_above = 0.0;
        auto enter = [&](int j) {   // exact totals of the bins above bin j
            n_above = (j + 1 < kLogNB) ? (unsigned long long)n_ge[j + 1] : 0ull;
            s_above = (j + 1 < kLogNB) ? s_ge[j + 1] : 0.0;
        };
        if (!bad) enter(jb);
        uint32_t par = 0u;   // alternating slots: a wave may write iteration k + 1's partials while another still reads k's
        uint32_t done = 0u;
        DPL_PROF_T(pt2);
        DPL_PROF_ADD(1, pt1, pt2);
        while (!done && !bad) {
            // values of bin jb above s: bit patterns in (bits(s), lower edge of bin jb + 1), i.e. d = u - bits(s) - 1 below
            // `span` (unsigned: anything at or below s wraps around).  Four VALU instructions per value — the count is a
            // population count of the compare mask on the scalar unit — and the mantissa sum follows from the sum of d.
            const uint32_t lo1 = __float_as_uint(s) + 1u;
            const uint32_t span = (((uint32_t)(jb + 1) + kLogKey0) << kLogShift) - lo1;
            uint32_t c = 0u;   // (wave-uniform)
            unsigned long long dsum = 0ull;
            uint32_t ds = 0u;   // per thread: at most 80 values below 2^17 between two wave sums
            auto in1 = [&](float f) {
                const uint32_t d = __float_as_uint(f) - lo1;
                const bool in = d < span;
                c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
                ds += in ? d : 0u;
            };
            if (n_rows > (uint32_t)kVecT) load_rows(ov, std::integral_constant<int, kOver>{}, (uint32_t)kVecT, L);
            {
                const uint32_t rows = min(n_rows, (uint32_t)kVecT);
#pragma unroll
                for (int u = 0; u < kVecT; ++u) {
                    if ((uint32_t)u < rows) {   // uniform
                        in1(v[u].x);
                        in1(v[u].y);
                        in1(v[u].z);
                        in1(v[u].w);
                    }
                }
                dsum += (unsigned long long)wave_sum_dpp(ds);   // < 64 * 80 * 2^17
                ds = 0u;
            }
            for (uint32_t r0 = (uint32_t)kVecT; r0 < n_rows; r0 += (uint32_t)kOver) {
#pragma unroll
                for (int u = 0; u < kOver; ++u) {   // (rows past the list's end were loaded as zeros)
                    in1(ov[u].x);
                    in1(ov[u].y);
                    in1(ov[u].z);
                    in1(ov[u].w);
                }
                if (r0 + (uint32_t)kOver < n_rows) load_rows(ov, std::integral_constant<int, kOver>{}, r0 + (uint32_t)kOver, L);
                dsum += (unsigned long long)wave_sum_dpp(ds);
                ds = 0u;
            }
            if (lane == 0) {
                sh.part_c[par][w] = c;
                sh.part_m[par][w] = dsum + (unsigned long long)c * (unsigned long long)(lo1 & 0x7FFFFFu);   // sum of explicit mantissas
            }
            __syncthreads();
            {
                unsigned long long tc = 0ull, tm = 0ull;
#pragma unroll
                for (int j = 0; j < kWaves; ++j) {
                    tc += sh.part_c[par][j];
                    tm += sh.part_m[par][j];
                }
                par ^= 1u;
                const unsigned long long tg = n_above + tc;
                const double ts = s_above + (double)(tm + (tc << 23)) * log_bin_scale(jb);
                const OctavStep qs = octav_step(ts, tg, n_elems - tg, ud, s, iters, max_iters);
                s = qs.s;
                iters = qs.iters;
                done = qs.done;
                if (!done) {
                    const int jn = log_bin(s);
                    if (!marked(jn)) {
                        bad = 1u;   // a bin that was not gathered (or out of the binned window): the compaction route takes over
                    } else if (jn != jb) {
                        jb = jn;
                        enter(jb);
                    }
                }
            }
        }
        DPL_PROF_T(pt3);
        DPL_PROF_ADD(2, pt2, pt3);
        if (tid == 0) g_prof_iters_add(blockIdx.x, iters);
    }
    if (tid == 0) {
        if (bad) {
            // restart from s_0 (in me->s) on the compaction route: state as k_octav_update<true> leaves it
            me->mode = 1u;
            me->done = 0u;
            me->len[0] = 0u;
            atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
        } else {
            me->s = s;
            me->iters = iters;
            me->done = 1u;
            me->mode = 2u;
        }
    }
}

#ifndef DPL_TAIL_LDS_PAD
#define DPL_TAIL_LDS_PAD 0   // (occupancy experiments: extra dynamic LDS per workgroup)
#endif
#include "octav_tail.hpp"

// The rescue walk: a small persistent grid over the list of rescued pairs — usually empty, and a launch that has nothing to do
// should not have thousands of workgroups to schedule between those of the next batch's streaming kernel.
__global__ __launch_bounds__(kThreads, DPL_WALK_OCC) void k_octav_walk_rescue(
    dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl, const uint64_t* __restrict__ pair_base, int max_iters,
    int fail_every, const uint32_t* __restrict__ rescue_bm, const uint32_t* __restrict__ missed,
    const float* __restrict__ list_rescue, const unsigned long long* __restrict__ resc) {
    __shared__ double s_ge[kLogNB];
    __shared__ uint32_t n_ge[kLogNB];
    __shared__ Shared sh;
    const uint32_t n_missed = ctl->len[0];
    for (uint32_t e = blockIdx.x; e < n_missed; e += gridDim.x) {
        walk_rescued<kVec>(missed[3 * e], s_ge, n_ge, sh, st, ctl, pair_base, max_iters, fail_every, rescue_bm, list_rescue, resc);
        __syncthreads();
    }
}

}  // namespace

extern int g_exact_fail_every, g_rescue_fail_every;   // octav_kernels.hip (dpl_test_hook_exact_fail_every / _rescue_fail_every)
int dpl_octav_rescue_gather_launch(const uint32_t* d_missed, dpl_octav_state* d_states, int64_t n_pairs, const dpl_span* d_pair_spans,
                                   const float* const* d_seg_ptrs, const uint32_t* d_bm_rows, const uint64_t* d_pair_base,
                                   float* d_list1, hipStream_t st);
int dpl_octav_fallback_route(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                             const float* const* d_seg_ptrs, dpl_octav_state* d_states, int64_t n_pairs,
                             const dpl_span* d_pair_spans, const uint64_t* d_pair_base, const uint32_t* d_pair_order,
                             float* d_list0, float* d_list1, int dynamic_sym, int max_iters, hipStream_t st);

extern "C" {

#ifdef DPL_RES_PROF
int dpl_res_prof_read(unsigned long long* host_out, int reset) {   // tuning builds only
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_res_prof), sizeof(unsigned long long) * 4096 * 8);
    if (e != hipSuccess) return fail("dpl_res_prof_read", e);
    if (reset) {
        static unsigned long long z[4096 * 8];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_res_prof), z, sizeof(z));
        if (e != hipSuccess) return fail("dpl_res_prof_read", e);
    }
    return 0;
}
#endif

uint32_t dpl_octav_slice_cap(void) { return kCap; }
uint32_t dpl_octav_list_cap(uint64_t n_elements) { return list_cap_of(n_elements); }
uint32_t dpl_octav_small_pair(void) { return kSmallCap; }

int64_t dpl_build_octav_slices(const dpl_span* spans, int64_t n_spans, dpl_work_item* out, int64_t cap, uint32_t* pair_slice0) {
    if (!spans || n_spans < 0) return fail_msg("dpl_build_octav_slices: bad arguments");
    // largest pairs first: the long ones start at once, the short ones fill the tail of the launch
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_spans > 0 ? n_spans : 1));
    if (!order) return fail_msg("dpl_build_octav_slices: out of memory");
    for (int64_t i = 0; i < n_spans; ++i) order[i] = i;
    struct Cmp {
        static int f(const void* a, const void* b, void* ctx) {
            const dpl_span* sp = (const dpl_span*)ctx;
            const int64_t ia = *(const int64_t*)a, ib = *(const int64_t*)b;
            if (sp[ia].count != sp[ib].count) return sp[ia].count > sp[ib].count ? -1 : 1;
            return ia < ib ? -1 : (ia > ib ? 1 : 0);
        }
    };
    qsort_r(order, (size_t)n_spans, sizeof(int64_t), Cmp::f, (void*)spans);
    int64_t n_total = 0;
    for (int64_t oi = 0; oi < n_spans; ++oi) {
        const dpl_span& sp = spans[order[oi]];
        const uint64_t c = sp.count == 0 ? 0 : (sp.count + kCap - 1) / kCap;
        if (c > kMaxCluster) {
            free(order);
            snprintf(g_err, sizeof(g_err), "dpl_build_octav_slices: a pair of %llu elements needs %llu slices (max %u)",
                     (unsigned long long)sp.count, (unsigned long long)c, kMaxCluster);
            return -3;
        }
        n_total += (int64_t)c;
    }
    if (out && n_total <= cap) {
        int64_t p = 0;
        // pair_slice0[2 slot], [2 slot + 1]: first and one-past-last slice of the pair in slot `slot` (slots 0 .. n_spans-1)
        if (pair_slice0)
            for (int64_t i = 0; i < 2 * n_spans; ++i) pair_slice0[i] = 0u;
        for (int64_t oi = 0; oi < n_spans; ++oi) {
            const dpl_span& sp = spans[order[oi]];
            if (sp.count == 0) continue;
            const uint64_t c = (sp.count + kCap - 1) / kCap;
            const uint64_t per = (((sp.count + c - 1) / c) + 3) & ~3ull;   // equal slices, cut on multiples of 4 elements
            if (pair_slice0 && sp.slot < (uint64_t)n_spans) {
                pair_slice0[2 * sp.slot] = (uint32_t)p;
                pair_slice0[2 * sp.slot + 1] = (uint32_t)(p + (int64_t)c);
            }
            uint64_t off = 0;
            for (uint64_t j = 0; j < c; ++j) {
                const uint64_t take = (j + 1 == c) ? sp.count - off : per;
                out[p].offset = sp.offset + off;
                out[p].count = (uint32_t)take;
                out[p].seg = sp.seg;
                out[p].slot = sp.slot;
                out[p].reserved = (uint32_t)c;
                ++p;
                off += take;
            }
        }
    }
    free(order);
    return n_total;
}

static int check_job(const char* who, const dpl_octav_oneread_job* j) {
    if (!j) return fail_msg("dpl_octav_oneread: null job");
    if (j->n_pairs <= 0 || j->n_slices <= 0) return 1;   // nothing to do
    if (j->n_tensors < 1 || (j->write_epoch != 0 && j->write_epoch != 1)) {
        snprintf(g_err, sizeof(g_err), "%s: bad tensor count / epoch", who);
        return -1;
    }
    if (!j->d_slices || !j->d_pair_slice0 || !j->d_pair_spans || !j->d_pair_base || !j->d_pair_order || !j->d_seg_ptrs ||
        !j->d_states || !j->d_pred || !j->d_list0 || !j->d_list1 || !j->d_rescue_bm || !j->d_missed || !j->d_vis || !j->d_resc ||
        (j->n_multi > 0 && !j->d_lh)) {
        snprintf(g_err, sizeof(g_err), "%s: null buffer in the job", who);
        return -1;
    }
    if (j->n_small < 0 || j->n_small > j->n_pairs || j->n_multi < 0 || j->n_multi > j->n_pairs) {
        snprintf(g_err, sizeof(g_err), "%s: bad small-pair / multi-slice-pair count", who);
        return -1;
    }
    return 0;
}
#define DPL_JOB_CHECK(who)                       \
    if (int e_ = check_job(who, j)) return e_ > 0 ? 0 : e_

// state of every pair + the control block, and the tensors' threshold snapshot (what their pairs asked for in the current and
// the previous epoch of batches)
int dpl_octav_oneread_prepare(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_prepare");
    const int64_t vis_words = j->n_tensors * kLogWords;
    uint32_t* d_vis_w = j->d_vis + (int64_t)j->write_epoch * vis_words;
    const uint32_t* d_vis_o = j->d_vis + (int64_t)(1 - j->write_epoch) * vis_words;
    const int64_t n = j->n_pairs + 1 > j->n_tensors ? j->n_pairs + 1 : j->n_tensors;
    hipLaunchKernelGGL(k_octav_tail_init, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, j->d_states, j->n_pairs, d_vis_w, d_vis_o,
                       j->d_pred, j->n_tensors, j->reset_epoch);
    DPL_LAUNCH_CHECK("k_octav_tail_init");
    return 0;
}

// the batch's only read of the activations: k_octav_tail streams AND walks every single-slice pair; the slices of a pair above one
// slice leave their rows for k_octav_tail_merge
int dpl_octav_oneread_stream(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_stream");
    const TailArgs fa{j->d_vis + (int64_t)j->write_epoch * j->n_tensors * kLogWords, j->d_pred, j->d_rescue_bm, j->d_missed,
                      reinterpret_cast<unsigned long long*>(j->d_resc), j->dynamic_sym, j->max_iters, g_exact_fail_every};
    const size_t lds = (size_t)(kLdsA + kTailLdsB + DPL_TAIL_LDS_PAD);
    // (a slice of a pair above one slice — the first items of d_slices, largest first — leaves its row in d_lh ...)
    hipLaunchKernelGGL(k_octav_tail, dim3((unsigned)j->n_slices), dim3(kThreads), lds, (hipStream_t)s, j->d_slices,
                       j->d_seg_ptrs, j->d_states, (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_states + j->n_pairs,
                       j->d_pair_spans, reinterpret_cast<unsigned long long*>(j->d_lh), fa);
    DPL_LAUNCH_CHECK("k_octav_tail");
    if (j->n_multi > 0) {   // ... which one workgroup per such pair adds up and walks (d_pair_order: these pairs come first)
        hipLaunchKernelGGL(k_octav_tail_merge, dim3((unsigned)j->n_multi), dim3(kThreads), lds, (hipStream_t)s, j->d_slices, j->d_states,
                           (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_states + j->n_pairs, j->d_pair_spans,
                           reinterpret_cast<const unsigned long long*>(j->d_lh), j->d_pair_order, j->d_pair_slice0, fa);
        DPL_LAUNCH_CHECK("k_octav_tail_merge");
    }
    return 0;
}

// Everything behind the streaming kernel, in stream order, nothing decided on the host: the RESCUE of the pairs whose walk was
// refused (k_octav_rescue_gather: those pairs re-read alone for their exact bracket's bins; k_octav_walk_rescue: the verified walk
// of every iterate), then — compaction_inline — the compaction route for what even that could not finish.  Both kernels return
// at once when the control block lists nothing (the usual case: 3 of 3 936 pairs of a ResNet-50 batch are rescued).
int dpl_octav_oneread_finish(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_finish");
    hipStream_t st = (hipStream_t)s;
    dpl_octav_state* ctl = j->d_states + j->n_pairs;
    if (j->max_iters <= 0) return 0;
    if (int e = dpl_octav_rescue_gather_launch(j->d_missed, j->d_states, j->n_pairs, j->d_pair_spans, j->d_seg_ptrs, j->d_rescue_bm,
                                               j->d_pair_base, j->d_list1, st))
        return e;
    hipLaunchKernelGGL(k_octav_walk_rescue, dim3(kRescueGrid), dim3(kThreads), 0, st, j->d_states, ctl, j->d_pair_base, j->max_iters,
                       g_rescue_fail_every, j->d_rescue_bm, j->d_missed, j->d_list1, reinterpret_cast<const unsigned long long*>(j->d_resc));
    DPL_LAUNCH_CHECK("k_octav_walk_rescue");
    return j->compaction_inline ? dpl_octav_oneread_compaction(j, s) : 0;
}

// The compaction route for the pairs the control block counts in cnt_le (after dpl_octav_oneread_finish): what neither the walk
// nor the rescue could finish.  Its kernels return at once when there is none, but a caller that can read the count later (the
// pipeline: two batches on) skips the call — four launches with large footprints would otherwise wait for slots beside the
// next batch's streaming kernel.
int dpl_octav_oneread_compaction(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_compaction");
    if (j->max_iters <= 0) return 0;
    if (int e = check_blocks("dpl_octav_oneread_compaction", j->n_items, j->d_block_begin, j->n_blocks)) return e;
    hipStream_t st = (hipStream_t)s;
    // its own whole-pair list regions (d_pair_base_full) in its own two lists: the one-read forms' lists hold list_cap_of(n)
    // values per pair, and the streaming kernel of a later batch may be writing d_list0 by now
    if (!j->d_pair_base_full || !j->d_clist0 || !j->d_clist1)
        return fail_msg("dpl_octav_oneread_compaction: the compaction route's lists (d_pair_base_full, d_clist0, d_clist1) are missing");
    return dpl_octav_fallback_route(j->d_items, j->n_items, j->d_block_begin, j->n_blocks, j->d_seg_ptrs, j->d_states, j->n_pairs,
                                    j->d_pair_spans, j->d_pair_base_full, j->d_pair_order, j->d_clist0, j->d_clist1, j->dynamic_sym,
                                    j->max_iters, st);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The exact-tail form as a SELF-SUFFICIENT ABI (forward_net.py:315-340 is the call site it serves): a HOST plan over the pairs of
// one tensor-set geometry knows every buffer's size, uploads the static tables and fills the job.  A caller allocates what
// dpl_octav_plan_sizes reports, nothing else.
struct dpl_octav_plan {
    int64_t n_pairs = 0, n_tensors = 0, n_slices = 0, n_multi = 0, n_small = 0, n_items = 0, n_blocks = 0, n_multi_slices = 0;
    uint64_t list_elems = 0, full_elems = 0;
    // host copies of the tables, in the order they sit in the device block (offsets below, bytes)
    dpl_work_item* slices = nullptr;
    uint32_t* pair_slice0 = nullptr;
    dpl_span* spans = nullptr;
    uint64_t* pair_base = nullptr;       // [n_pairs + 1]: capped regions
    uint64_t* pair_base_full = nullptr;  // [n_pairs + 1]: whole-pair regions (the compaction route's lists)
    uint32_t* pair_order = nullptr;
    dpl_work_item* items = nullptr;
    uint32_t* block_begin = nullptr;
    uint64_t off_slices = 0, off_ps0 = 0, off_spans = 0, off_base = 0, off_basef = 0, off_order = 0, off_items = 0, off_bb = 0, tables = 0;
};
static uint64_t up256(uint64_t x) { return (x + 255ull) & ~255ull; }

dpl_octav_plan* dpl_octav_plan_create(const dpl_span* spans, int64_t n_spans, int64_t n_tensors, int64_t n_blocks) {
    if (!spans || n_spans < 1 || n_tensors < 1 || n_blocks < 1) {
        fail_msg("dpl_octav_plan_create: bad arguments");
        return nullptr;
    }
    for (int64_t i = 0; i < n_spans; ++i)
        if (spans[i].slot != (uint32_t)i) {
            fail_msg("dpl_octav_plan_create: spans must carry slots 0 .. n_spans-1 in order (slot = image * n_tensors + tensor)");
            return nullptr;
        }
    dpl_octav_plan* p = new dpl_octav_plan();
    p->n_pairs = n_spans;
    p->n_tensors = n_tensors;
    p->n_blocks = n_blocks;
    const int64_t ns = dpl_build_octav_slices(spans, n_spans, nullptr, 0, nullptr);
    if (ns < 0) {   // (-3: a pair above 64 slices: dpl_octav_run_bracket serves such a set)
        delete p;
        return nullptr;
    }
    p->n_slices = ns;
    p->slices = (dpl_work_item*)calloc((size_t)(ns > 0 ? ns : 1), sizeof(dpl_work_item));
    p->pair_slice0 = (uint32_t*)calloc((size_t)(2 * n_spans), sizeof(uint32_t));
    p->spans = (dpl_span*)malloc(sizeof(dpl_span) * (size_t)n_spans);
    p->pair_base = (uint64_t*)calloc((size_t)(n_spans + 1), sizeof(uint64_t));
    p->pair_base_full = (uint64_t*)calloc((size_t)(n_spans + 1), sizeof(uint64_t));
    p->pair_order = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_spans);
    memcpy(p->spans, spans, sizeof(dpl_span) * (size_t)n_spans);
    dpl_build_octav_slices(spans, n_spans, p->slices, ns, p->pair_slice0);
    // list regions, in pair order: a single-slice pair list_cap_of(n) values, a pair of c slices c parts of list_cap_of(slice)
    for (int64_t i = 0; i < n_spans; ++i) {
        const uint64_t n = spans[i].count, c = n == 0 ? 0 : (n + kCap - 1) / kCap;
        uint64_t region = 0;
        if (c == 1) region = list_cap_of(n);
        else if (c > 1) region = c * (uint64_t)list_cap_of((((n + c - 1) / c) + 3) & ~3ull);
        p->pair_base[i + 1] = p->pair_base[i] + region;
        p->pair_base_full[i + 1] = p->pair_base_full[i] + ((n + 31ull) & ~31ull);
        if (c > 1) p->n_multi += 1, p->n_multi_slices += (int64_t)c;
        if (n <= kSmallCap) p->n_small += 1;
    }
    p->list_elems = p->pair_base[n_spans];
    p->full_elems = p->pair_base_full[n_spans];
    // pair order: largest first (stable) — the multi-slice pairs are its first n_multi entries, the small pairs its last n_small
    for (int64_t i = 0; i < n_spans; ++i) p->pair_order[i] = (uint32_t)i;
    struct Cmp {
        static int f(const void* a, const void* b, void* ctx) {
            const dpl_span* sp = (const dpl_span*)ctx;
            const uint32_t ia = *(const uint32_t*)a, ib = *(const uint32_t*)b;
            if (sp[ia].count != sp[ib].count) return sp[ia].count > sp[ib].count ? -1 : 1;
            return ia < ib ? -1 : (ia > ib ? 1 : 0);
        }
    };
    qsort_r(p->pair_order, (size_t)n_spans, sizeof(uint32_t), Cmp::f, (void*)spans);
    // the balanced partition of the same pairs (the compaction route's kernels)
    const int64_t ni = dpl_build_balanced_items(spans, n_spans, n_blocks, nullptr, 0, nullptr);
    if (ni < 0) {
        dpl_octav_plan_destroy(p);
        return nullptr;
    }
    p->n_items = ni;
    p->items = (dpl_work_item*)calloc((size_t)(ni > 0 ? ni : 1), sizeof(dpl_work_item));
    p->block_begin = (uint32_t*)calloc((size_t)(n_blocks + 1), sizeof(uint32_t));
    dpl_build_balanced_items(spans, n_spans, n_blocks, p->items, ni, p->block_begin);
    uint64_t o = 0;
    p->off_slices = o, o += up256(sizeof(dpl_work_item) * (uint64_t)(ns > 0 ? ns : 1));
    p->off_ps0 = o, o += up256(sizeof(uint32_t) * 2ull * (uint64_t)n_spans);
    p->off_spans = o, o += up256(sizeof(dpl_span) * (uint64_t)n_spans);
    p->off_base = o, o += up256(sizeof(uint64_t) * (uint64_t)(n_spans + 1));
    p->off_basef = o, o += up256(sizeof(uint64_t) * (uint64_t)(n_spans + 1));
    p->off_order = o, o += up256(sizeof(uint32_t) * (uint64_t)n_spans);
    p->off_items = o, o += up256(sizeof(dpl_work_item) * (uint64_t)(ni > 0 ? ni : 1));
    p->off_bb = o, o += up256(sizeof(uint32_t) * (uint64_t)(n_blocks + 1));
    p->tables = o;
    return p;
}

void dpl_octav_plan_destroy(dpl_octav_plan* p) {
    if (!p) return;
    free(p->slices);
    free(p->pair_slice0);
    free(p->spans);
    free(p->pair_base);
    free(p->pair_base_full);
    free(p->pair_order);
    free(p->items);
    free(p->block_begin);
    delete p;
}

// the layout of the per-batch blocks (bytes from their base)
static uint64_t state_pred_off(const dpl_octav_plan* p) { return up256(sizeof(dpl_octav_state) * (uint64_t)(p->n_pairs + 1)); }
static uint64_t rescue_missed_off(const dpl_octav_plan* p) { return up256(sizeof(uint32_t) * (uint64_t)kLogWords * (uint64_t)p->n_pairs); }
static uint64_t rescue_resc_off(const dpl_octav_plan* p) { return rescue_missed_off(p) + up256(sizeof(uint32_t) * 3ull * (uint64_t)p->n_pairs); }
static uint64_t rescue_lh_off(const dpl_octav_plan* p) { return rescue_resc_off(p) + up256(8ull * (uint64_t)kRescRow * (uint64_t)p->n_pairs); }

int dpl_octav_plan_sizes(const dpl_octav_plan* p, dpl_octav_workspace_sizes* out) {
    if (!p || !out) return fail_msg("dpl_octav_plan_sizes: null argument");
    out->tables_bytes = p->tables;
    out->history_bytes = sizeof(uint32_t) * 2ull * (uint64_t)p->n_tensors * (uint64_t)kLogWords;
    out->state_bytes = state_pred_off(p) + up256(sizeof(uint32_t) * (uint64_t)kPredRow * (uint64_t)p->n_tensors);
    out->rescue_bytes = rescue_lh_off(p) + 8ull * (uint64_t)kLogNB * (uint64_t)p->n_multi_slices;
    out->list_bytes = 4ull * (p->list_elems > 0 ? p->list_elems : 32ull);
    out->fallback_bytes = 2ull * 4ull * (p->full_elems > 0 ? p->full_elems : 32ull);
    out->result_bytes = sizeof(float) * 3ull * (uint64_t)p->n_pairs;
    out->n_pairs = p->n_pairs;
    out->n_slices = p->n_slices;
    out->n_multi = p->n_multi;
    out->n_small = p->n_small;
    return 0;
}

int dpl_octav_plan_upload(const dpl_octav_plan* p, void* d_tables, dpl_stream_t s) {
    if (!p || !d_tables) return fail_msg("dpl_octav_plan_upload: null argument");
    char* d = (char*)d_tables;
    hipStream_t st = (hipStream_t)s;
    const struct { uint64_t off; const void* src; uint64_t bytes; } parts[] = {
        {p->off_slices, p->slices, sizeof(dpl_work_item) * (uint64_t)p->n_slices},
        {p->off_ps0, p->pair_slice0, sizeof(uint32_t) * 2ull * (uint64_t)p->n_pairs},
        {p->off_spans, p->spans, sizeof(dpl_span) * (uint64_t)p->n_pairs},
        {p->off_base, p->pair_base, sizeof(uint64_t) * (uint64_t)(p->n_pairs + 1)},
        {p->off_basef, p->pair_base_full, sizeof(uint64_t) * (uint64_t)(p->n_pairs + 1)},
        {p->off_order, p->pair_order, sizeof(uint32_t) * (uint64_t)p->n_pairs},
        {p->off_items, p->items, sizeof(dpl_work_item) * (uint64_t)p->n_items},
        {p->off_bb, p->block_begin, sizeof(uint32_t) * (uint64_t)(p->n_blocks + 1)},
    };
    for (const auto& q : parts) {
        if (q.bytes == 0) continue;
        const hipError_t e = hipMemcpyAsync(d + q.off, q.src, q.bytes, hipMemcpyHostToDevice, st);   // (the plan owns the sources)
        if (e != hipSuccess) return fail("dpl_octav_plan_upload", e);
    }
    return 0;
}

int dpl_octav_plan_bind(const dpl_octav_plan* p, void* d_tables, void* d_history, void* d_state, void* d_rescue, void* d_list0,
                        void* d_list1, void* d_fallback, const float* const* d_seg_ptrs, int64_t call_index, int dynamic_sym,
                        int max_iters, dpl_octav_oneread_job* j) {
    if (!p || !j || !d_tables || !d_history || !d_state || !d_rescue || !d_list0 || !d_list1)
        return fail_msg("dpl_octav_plan_bind: null argument");
    if (call_index < 0) return fail_msg("dpl_octav_plan_bind: negative call index");
    memset(j, 0, sizeof(*j));
    char* t = (char*)d_tables;
    j->d_slices = (const dpl_work_item*)(t + p->off_slices);
    j->n_slices = p->n_slices;
    j->d_pair_slice0 = (const uint32_t*)(t + p->off_ps0);
    j->d_pair_spans = (const dpl_span*)(t + p->off_spans);
    j->d_pair_base = (const uint64_t*)(t + p->off_base);
    j->d_pair_base_full = (const uint64_t*)(t + p->off_basef);
    j->d_pair_order = (const uint32_t*)(t + p->off_order);
    j->n_pairs = p->n_pairs;
    j->n_tensors = p->n_tensors;
    j->n_small = p->n_small;
    j->n_multi = p->n_multi;
    j->d_items = (const dpl_work_item*)(t + p->off_items);
    j->n_items = p->n_items;
    j->d_block_begin = (const uint32_t*)(t + p->off_bb);
    j->n_blocks = p->n_blocks;
    j->d_seg_ptrs = d_seg_ptrs;
    j->d_states = (dpl_octav_state*)d_state;
    j->d_pred = (uint32_t*)((char*)d_state + state_pred_off(p));
    char* r = (char*)d_rescue;
    j->d_rescue_bm = (uint32_t*)r;
    j->d_missed = (uint32_t*)(r + rescue_missed_off(p));
    j->d_resc = (uint64_t*)(r + rescue_resc_off(p));
    j->d_lh = (uint64_t*)(r + rescue_lh_off(p));
    j->d_list0 = (float*)d_list0;
    j->d_list1 = (float*)d_list1;
    if (d_fallback) {
        j->d_clist0 = (float*)d_fallback;
        j->d_clist1 = (float*)d_fallback + (p->full_elems > 0 ? p->full_elems : 32ull);
    }
    j->d_vis = (uint32_t*)d_history;
    // two alternating epoch accumulators of kTailEpoch batches each: batch k adds to accumulator (k / epoch) % 2, cleared by the
    // first batch of an epoch (dpl_octav_oneread_prepare)
    j->write_epoch = (int32_t)((call_index / kPlanEpoch) % 2);
    j->reset_epoch = (call_index % kPlanEpoch) == 0 ? 1 : 0;
    j->dynamic_sym = dynamic_sym;
    j->max_iters = max_iters;
    j->compaction_inline = d_fallback ? 1 : 0;
    return 0;
}

// HOST: where the compaction route's lists hold the pairs that are still unfinished after dpl_octav_oneread_finish — whole-pair
// regions for those pairs (mode 1, not done), empty ones for every other pair: a caller that reads the states back when the
// control block reports such pairs allocates two lists of the returned size instead of two whole-batch ones (14 pairs of a cold
// ResNet-50 batch of 3 936: 40 MB instead of 6.8 GB).
int64_t dpl_octav_fallback_layout(const dpl_octav_state* h_states, int64_t n_pairs, uint64_t* h_base_out) {
    if (!h_states || !h_base_out || n_pairs < 0) return fail_msg("dpl_octav_fallback_layout: bad arguments");
    uint64_t at = 0;
    for (int64_t i = 0; i < n_pairs; ++i) {
        h_base_out[i] = at;
        if (h_states[i].mode == 1u && !h_states[i].done) at += (h_states[i].n_elems + 31ull) & ~31ull;
    }
    h_base_out[n_pairs] = at;
    return (int64_t)at;
}

int dpl_octav_run_oneread(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    if (int e = dpl_octav_oneread_prepare(j, s)) return e;
    if (int e = dpl_octav_oneread_stream(j, s)) return e;
    if (int e = dpl_octav_oneread_finish(j, s)) return e;
    return j && !j->compaction_inline ? dpl_octav_oneread_compaction(j, s) : 0;
}

}  // extern "C"
