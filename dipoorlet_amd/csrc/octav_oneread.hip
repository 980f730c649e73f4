// OCTAV ('-A mse', forward_net.py:284-342) in ONE read of the activations.
//
// Why not two reads: measured on MI355X (scripts/mall_probe.hip, profiles/r02/mall_probe.txt) a re-read of recently
// streamed data costs the same whether HBM or the 256 MiB Infinity Cache serves it (6.1-6.9 TB/s either way, one
// shared fabric), so the two-read bracket form of octav_kernels.hip cannot pass ~40 % of the roofline.  And a form that
// keeps a pair on chip until a leader has walked its bracket (tried first, round 2) spends its time waiting — three
// cross-workgroup round trips per slice against ~16 us of residency a CU can afford at HBM rate.
//
// So nothing waits here.  The values the exact iteration needs are the ones in the histogram bins its iterates fall
// into; WHICH bins is predicted — per tensor and batch either from the previous batches (the bins the same tensor's iterates
// visited, OR-ed over the images of two alternating epochs of batches, neighbours that hold next to nothing and the sparse
// tail added: the narrowest prediction while the images are alike) or from a strided sample of the pair itself (k_octav_probe:
// wider, costs a read of 1/16 of the pair, indifferent to how the images differ) — and every iterate of the exact walk is
// VERIFIED against the set that was gathered.  A pair whose iterate leaves the gathered bins is RESCUED on the device: its
// exact bracket from the histogram, a re-read of that pair alone by many workgroups, a second walk.  Results are the
// reference's iterate sequence either way; only the speed depends on the prediction.
//
// Kernels:
//   k_octav_oneread      one 256-thread workgroup per SLICE (<= kCap elements of one (image, tensor) pair): the slice's
//                        only HBM read; per element min / max and ONE returning 64-bit LDS add on the bin's word of an
//                        exact log-scale histogram (64 bins per octave: count + integer mantissa sum) whose bit 63 says
//                        "gather"; gathered values -> dense per-wave LDS queues -> the slice's own region of the pair's
//                        list (LDS cursor, no global atomic).  A slice that is a whole pair (kCap = 1 044 480: every pair
//                        of the ResNet-50 / ViT-B/16 sets) is then WALKED by the same workgroup (walk_pair, phase 3):
//                        histogram converted in place into suffix totals, list read back from L2.  Otherwise the
//                        histogram leaves as one row per slice for k_octav_walk.
//   walk_pair            the exact walk of one pair: suffix totals -> s_0 -> the reference's iteration with the list in
//                        registers (totals of the bins above the iterate's bin: exact integers; listed values of that
//                        bin: integer mantissa sums), every iterate verified; records the bins stepped into for the next
//                        batches and what the tensor's prediction from earlier batches would have cost (the choice);
//                        a pair that leaves the gathered set: its bracket + suffix totals for the rescue.
//   k_octav_walk<16|32>  walk_pair for multi-slice pairs (one workgroup per pair); k_octav_walk_rescue: a persistent grid
//                        over the rescued pairs (octav_kernels.hip: k_octav_rescue_gather re-reads them first).
//   k_octav_sort + k_octav_walk_sorted   the walk for LONG lists of multi-slice pairs: a slice's list sorted by bin rank in
//                        8192-value runs, then one WAVE per pair looking up one rank per iteration.
//   k_octav_probe        the pair's own prediction row from a strided sample (128 bytes of every 2 KiB).
//   k_octav_oneread_init state + the tensors' prediction snapshot (bitmap of at most 255 bins + per-word rank prefix) of a
//                        batch + the choice per tensor.
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
#ifndef DPL_RES_OCC
#define DPL_RES_OCC 4
#endif
#ifndef DPL_CHEAP_SHIFT
#define DPL_CHEAP_SHIFT 11  // a neighbour of a visited bin is gathered too when it holds at most n >> 11 of the pair's n elements
#endif
#ifndef DPL_THIN_SHIFT
#define DPL_THIN_SHIFT 9    // ... and so is every bin above which at most n >> 9 elements lie (measured: 7 / 9 -> 9 / 11: +3 %, no more misses)
#endif
#ifndef DPL_SORTED_OCC
#define DPL_SORTED_OCC 4   // waves per SIMD of k_octav_walk_sorted: every pair of a batch resident at once (15 per CU at 32 x 123 pairs)
#endif
#ifndef DPL_WALK_OCC
#define DPL_WALK_OCC 4
#endif
constexpr int kVec = DPL_RES_VEC;                               // 16-byte vectors per thread the walk keeps a list in
constexpr int kOver = 4;                                         // rows of a list beyond the resident ones streamed per step of an iteration
constexpr uint32_t kSmallCap = 20480;                           // pairs this small gather their whole window (no prediction)
static_assert(kSmallCap == kListWhole, "a small pair's list region must hold the whole pair (list_cap_of)");
#ifndef DPL_SLICE_CAP
#define DPL_SLICE_CAP 1044480
#endif
constexpr uint32_t kCap = DPL_SLICE_CAP;                        // elements of a slice (streamed tile by tile)
static_assert(kCap < (1u << 20) && kCap % 4096u == 0u, "a slice's bin counts must fit the packed field below the flag bit");
#ifndef DPL_QUEUE_CAP
#define DPL_QUEUE_CAP 832
#endif
constexpr int kQueueCap = DPL_QUEUE_CAP;                        // entries of a WAVE's dense survivor queue; flushed above cap - 256
static_assert(kQueueCap > 256, "a vector of four elements per lane may add 256 survivors");
#ifndef DPL_QUEUE_TOP
#define DPL_QUEUE_TOP 256
#endif
constexpr int kQueueTop = DPL_QUEUE_TOP;                        // ... and, at a tile's drain point, above this many entries
#ifndef DPL_APPEND_LAG
#define DPL_APPEND_LAG 1
#endif
constexpr int kAppendLag = DPL_APPEND_LAG;                      // vectors between a vector's adds and the look at their returns
constexpr uint32_t kMaxCluster = 64;
constexpr int64_t kPlanEpoch = 8;   // batches per threshold-history epoch (dpl_octav_plan_bind; the Python pipeline's _ONEREAD_EPOCH)
#ifndef DPL_RESCUE_GRID
#define DPL_RESCUE_GRID 512
#endif
constexpr unsigned kRescueGrid = DPL_RESCUE_GRID;   // workgroups of the rescue's persistent kernels
// The prediction row of a tensor (d_pred): the bitmap of the bins to gather (kLogWords words, at most kMaxFlag bits set) and,
// per word, the number of set bits in the words below it — the RANK of a gathered bin is a table index everywhere below.
constexpr int kMaxFlag = 256;
constexpr int kPredRow = 2 * kLogWords;
// Sorted runs (k_octav_sort): a slice's list is sorted, kChunk values at a time, by the rank of the values' bins; the
// directory row of a chunk holds, per rank, the position of the rank's first value in the chunk (+ the chunk's length).
constexpr uint32_t kChunk = 8192;
constexpr int kDirRow = kMaxFlag + 8;   // uint16 entries; a multiple of 8: rows are 16-byte aligned
// Per tensor, what its prediction from earlier batches would have cost lately (floats, halved every batch):
// [0] values it would have listed, [1] elements walked, [2] walks it would not have covered, [3] walks, [4] current choice,
// [5] / [6] values listed / elements walked while the tensor's pairs predicted from their own sample, [7] that share, remembered,
// [8] the tensor's bracket width z (0: the default), [9] / [10] walks that left the sample's bins / walks on the sample,
// [11] != 0: the tensor's pairs are sampled at twice the default rate
constexpr int kTstatRow = 12;
constexpr int kRescRow = kLogNB + kLogNB / 2;   // u64 words of a rescued pair's row: 2048 suffix sums (fp64) + 2048 suffix counts (u32)
#ifndef DPL_PROBE_RATE
#define DPL_PROBE_RATE 16
#endif
constexpr uint32_t kProbeRate = DPL_PROBE_RATE;   // k_octav_probe reads one 128-byte chunk of every kProbeRate (or kProbeRate / 2)
constexpr float kProbeZ = 3.0f;                  // default width of the sample's brackets, in standard deviations
constexpr uint32_t kProbeThin = 64;               // sampled values above a bracket's lower end below which the whole tail is gathered

// LDS: [A: packed histogram 16 KiB, bit 63 of a word = gather flag | one dummy word per lane][B: the waves' survivor queues 13 KiB]
constexpr int kLdsA = kLogNB * 8 + kWave * 8;                   // + the lanes' dummy words
constexpr int kLdsB = kWaves * kQueueCap * 4;
static_assert(kLdsB >= kLogNB * 4, "the fused walk keeps its suffix counts in the queues' space");

template <class T>
__device__ __forceinline__ T ld_agent(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_agent(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ T add_agent(T* p, T v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef DPL_RES_PROF
// phase cycle counters of a tuning build (scripts/res_prof.py): [workgroup][8] u64, accumulated by thread 0
__device__ unsigned long long g_res_prof[4096 * 8];
#define DPL_PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define DPL_PROF_ADD(slot, a, b) do { if (threadIdx.x == 0) g_res_prof[(blockIdx.x & 4095u) * 8 + (slot)] += (b) - (a); } while (0)
#define DPL_PROF_WAVE(idx, slot, a, b) do { if ((threadIdx.x & 63u) == 0) g_res_prof[((idx) & 4095u) * 8 + (slot)] += (b) - (a); } while (0)
__device__ __forceinline__ void g_prof_iters_add(uint32_t b, uint32_t it) { g_res_prof[(b & 4095u) * 8 + 7] += it; }
#define DPL_PROF_L(len) g_res_prof[(blockIdx.x & 4095u) * 8 + 6] += (len)
#else
#define DPL_PROF_L(len) do {} while (0)
__device__ __forceinline__ void g_prof_iters_add(uint32_t, uint32_t) {}
#define DPL_PROF_T(var) do {} while (0)
#define DPL_PROF_ADD(slot, a, b) do {} while (0)
#define DPL_PROF_WAVE(idx, slot, a, b) do {} while (0)
#endif

// Where a pair's prediction row lives: the tensor's row from earlier batches, or the pair's own row from k_octav_probe —
// by this batch's choice for the tensor.
struct PredRows {
    const uint32_t* t;     // [n_tensors, kPredRow]
    const uint32_t* p;     // [n_pairs, kPredRow]
    const uint32_t* use;   // [n_tensors]
    __device__ __forceinline__ const uint32_t* row(uint32_t pair, uint32_t tensor) const {
        return use[tensor] ? p + (uint64_t)pair * kPredRow : t + (uint64_t)tensor * kPredRow;
    }
};

struct Shared {
    double red_d[kWaves];
    unsigned long long red_q[kWaves];
    uint32_t red_a[kWaves], red_b[kWaves];
    unsigned long long part_m[2][kWaves];   // walk: the waves' partial (count, mantissa sum), two alternating slots
    uint32_t part_c[2][kWaves];
    float red_mn[kWaves], red_mx[kWaves];
    double low_sum;               // streaming kernel: non-zero values outside the window: their sum, count, a NaN among them
    uint32_t low_cnt, low_nan;
    uint32_t bm[kLogWords];       // walker: the gathered bins, then this pair's bracket
    uint32_t pub[kLogWords];      // bins published for the next batch (bracket + cheap neighbours)
    uint32_t cheapw[kLogWords], thinw[kLogWords];   // walk: bins holding next to nothing / the sparse tail (bitmaps)
    uint32_t would_list;          // walk: values the tensor's prediction from earlier batches would have listed of this pair
    uint32_t cursor;              // streaming kernel: entries of the slice's list region handed out so far
    uint32_t list_cap, region_cap;   // exact-tail form: values this workgroup's list part / the pair's whole list region holds
    uint32_t tail_j;              // exact-tail form (octav_tail.hpp): the bin at and above which values are listed (only ever raised)
    uint32_t jwant;               // ... and the bin this pair asks the tensor's next batches to list from
    // ... wave 0 walks alone; what it hands to the others (and to the pair's state) at the joints of the walk
    float t_s, w_s0, w_ud;
    int w_jb;
    uint32_t w_evals, w_exact, w_path, w_bad, w_route, w_lkn, w_lkc;
    double w_lks;
    double f_sum;                 // streaming kernel -> its own walk (a single-slice pair): the statistics it just published
    uint32_t f_nz, f_nan;
    float f_mn, f_mx;
    uint32_t seg_off[kMaxCluster], seg_len[kMaxCluster];   // walk: the pair's list segments (one per slice)
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

// wave64 inclusive prefix sum by DPP (Hillis-Steele inside each row of 16, then the two row broadcasts): VALU only — the
// ds_bpermute form is six dependent LDS round trips
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2, 3
    return v;
}

// One slice, streamed: per element min / max, the LDS histogram, and a queue append for the values of marked bins
// (queues -> the slice's own region of the pair's list; the region's cursor lives in LDS).  Leaves the per-wave statistics
// in sh.red_*.
// Its own function (not inlined): the register allocator otherwise spills the tile buffers of this hot loop to make
// room for values that only the walk needs.  The dynamic LDS block is addressed through address-space-3 pointers taken
// here (not handed in): ds_ instructions with constant offsets, nothing reloaded.
typedef __attribute__((address_space(3))) unsigned long long* lptr_u64;
typedef __attribute__((address_space(3))) uint32_t* lptr_u32;
#ifdef DPL_WITH_ONEREAD   // the round-3 form's streaming pass
__device__ __attribute__((noinline)) void stream_slice(const float* __restrict__ pg, uint32_t cnt, uint32_t* __restrict__ dst,
                                                       Shared& sh, dpl_octav_state* __restrict__ ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const lptr_u64 l_packed = (lptr_u64)(lds_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    float mn = INFINITY, mx = -INFINITY;
    // the wave's survivor queue: dense (ballot + mbcnt positions), `tail` entries in use (wave-uniform)
    const lptr_u32 wq = (lptr_u32)(lds_raw + kLdsA) + (uint32_t)w * kQueueCap;
    uint32_t tail = 0u;
    // A flush touches no global atomic: the slice owns the part of the pair's list that starts at the slice's offset in the
    // pair (as many entries as the slice has elements), and the position inside it comes from a cursor in LDS.  Flushes
    // need care: gfx950 has ONE in-order counter for loads and stores (vmcnt), and the waits for tile data are counts the
    // compiler fixes statically — a store issued between a tile's loads and the wait for them makes that wait also wait
    // for the store's acknowledgement (measured, round 2: per-lane queues flushed in mid-tile every ~7 tiles cost 56 us
    // of 634; dropping a returning global atomic from the flush changed nothing).  So the regular flush (queue more than
    // kQueueTop full) happens at the START of a tile's consumption, once the whole tile has arrived: the only loads
    // outstanding then are the next tile's, and by the time those are waited for — a tile's worth of work later — the
    // stores have long been acknowledged.  A flush in the middle of a tile remains for the case that one tile brings more
    // than the rest of the queue holds (a small pair gathering its whole window).  The queue is dense, so the stores are
    // full 256-byte instructions.
    auto flush = [&]() {
        typedef __attribute__((address_space(1))) uint32_t* gptr_u32;   // global, not flat: see for_each_tile
        gptr_u32 gdst = (gptr_u32)dst;
        uint32_t base = 0u;
        if (lane == 0) base = __hip_atomic_fetch_add((lptr_u32)&sh.cursor, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t i = lane; i < tail; i += kWave) gdst[base + i] = wq[i] & 0x7FFFFFFFu;   // |x|; the walk is a later launch
        tail = 0u;
    };
    // Four elements at a time, straight-line and branch-free.  ONE LDS operation per element does both jobs: the histogram
    // word of a bin the walk is predicted to visit carries a flag in bit 63 (set when the workgroup initialises its
    // histogram), and the RETURNING 64-bit add {count += 1, mantissa sum += 23 explicit bits} hands the flag back — there is
    // no separate look-up of the predicted set (measured, round 2: that look-up — address, ds_read, bit extract, ~7 of ~21
    // VALU instructions per element — cost 95 us of a 680 us kernel that is bound by instruction issue; the atomics
    // themselves 4 us).  A zero or a value outside the window adds to a per-LANE dummy word behind the histogram instead of
    // being masked off (no exec juggling, no same-address pile-up: lane l's dummy lies in bank pair l); its flag is never
    // set.  A non-zero value outside the window (or NaN) leaves a per-lane mark that is looked at once per tile.
    // `issue` starts the four adds of a vector; `append` (a vector later: the returns have arrived by then) puts the flagged
    // elements at the wave's queue tail (position = tail + the number of flagged lanes below: ballot + v_mbcnt).
    uint32_t rare = 0u;
    struct Ret4 {
        uint32_t h[4];   // high words of the returned histogram entries (bit 31 = the flag)
    };
    constexpr uint32_t kWin = (uint32_t)(kLogNB - 1);   // window bins 1 .. kLogNB-1 (as LogHistOp, octav_kernels.hip)
    const lptr_u64 dummy = l_packed + kLogNB + lane;
    auto add1 = [&](uint32_t bits) {
        const uint32_t t = ((bits >> kLogShift) & 0x3FFFu) - (kLogKey0 + 1u);   // key = 14 bits of exponent and top mantissa
        const bool in = t < kWin;
        const lptr_u64 slot = in ? l_packed + t + 1u : dummy;
        rare |= in ? 0u : bits;
        return (uint32_t)(__hip_atomic_fetch_add(slot, (1ull << kPackShift) | (unsigned long long)(bits & 0x7FFFFFu), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
    };
    auto issue = [&](const f4& t4) {
        Ret4 r;
        r.h[0] = add1(__float_as_uint(t4.x));
        r.h[1] = add1(__float_as_uint(t4.y));
        r.h[2] = add1(__float_as_uint(t4.z));
        r.h[3] = add1(__float_as_uint(t4.w));
        return r;
    };
    auto put = [&](uint32_t bits, uint32_t hi) {
        const bool f = (int32_t)hi < 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail));
        if (f) wq[pos] = bits;
        tail += (uint32_t)__builtin_popcountll(m);
    };
    auto append = [&](const f4& t4, const Ret4& r) {
#if defined(DPL_ABL_NOAPPEND)      // ablation builds (timing only): the returns are not looked at, nothing is queued
#else
        put(__float_as_uint(t4.x), r.h[0]);
        put(__float_as_uint(t4.y), r.h[1]);
        put(__float_as_uint(t4.z), r.h[2]);
        put(__float_as_uint(t4.w), r.h[3]);
        if (tail > (uint32_t)(kQueueCap - 256)) flush();   // (rare: see flush)
#endif
    };
    uint32_t rare_n = 0u;
    for_each_tile<kThreads>(pg, cnt, [&](const f4 (&t)[4], uint32_t base, bool full) {
        if (tail > (uint32_t)kQueueTop) {   // the regular flush: BEFORE the tile is consumed, AFTER all of it has arrived
            asm volatile("" ::"v"(t[3].w));   // (a use of the tile's last register: the compiler waits for the whole tile here)
#if defined(DPL_ABL_NOFLUSH)       // ablation builds (timing only): a full queue is simply dropped
            tail = 0u;
#else
            flush();
#endif
        }
        Ret4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (full) {
                mn = fminf(mn, fminf(fminf(t[u].x, t[u].y), fminf(t[u].z, t[u].w)));
                mx = fmaxf(mx, fmaxf(fmaxf(t[u].x, t[u].y), fmaxf(t[u].z, t[u].w)));
            } else {   // padding is +0.0: in no histogram bin, in no marked bin; only min / max must skip it
                const uint32_t e = base + (uint32_t)u * 256u + lane * 4u;
                if (e + 0 < cnt) mn = fminf(mn, t[u].x), mx = fmaxf(mx, t[u].x);
                if (e + 1 < cnt) mn = fminf(mn, t[u].y), mx = fmaxf(mx, t[u].y);
                if (e + 2 < cnt) mn = fminf(mn, t[u].z), mx = fmaxf(mx, t[u].z);
                if (e + 3 < cnt) mn = fminf(mn, t[u].w), mx = fmaxf(mx, t[u].w);
            }
            r[u] = issue(t[u]);
            if (u >= kAppendLag) append(t[u - kAppendLag], r[u - kAppendLag]);
        }
#pragma unroll
        for (int u = 4 - kAppendLag; u < 4; ++u) append(t[u], r[u]);
        if (__any((rare & 0x7FFFFFFFu) != 0u)) {   // (a lone sign bit is -0.0: nothing to account for)
            // Non-zero values outside the window (and NaNs) are accumulated directly — from the tile's registers, behind this
            // wave-uniform branch, into three words of LDS (one set of atomics per wave and tile): rare on convolutional
            // activations, every tile of an attention-probability tensor (round 3: such tiles used to be noted and read a
            // second time after the slice; 5.8 % of the tiles of the ViT-B/16 set)
            double fs = 0.0;
            uint32_t c = 0u, nn = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t a = __float_as_uint(xs[e]) & 0x7FFFFFFFu;
                    const uint32_t tt = (a >> kLogShift) - (kLogKey0 + 1u);
                    const bool o = !(tt < (uint32_t)(kLogNB - 1)) && a != 0u;
                    const float f = __uint_as_float(a);
                    const bool pos = o && f > 0.0f;
                    fs += pos ? (double)f : 0.0;
                    c += pos ? 1u : 0u;
                    nn |= (o && f != f) ? 1u : 0u;
                }
            }
            const uint32_t ct = (uint32_t)__builtin_amdgcn_readlane((int)scan_u32_dpp(c), kWave - 1);
            const unsigned long long fb = (unsigned long long)__double_as_longlong(scan_f64_dpp(fs));
            const double ft = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(fb >> 32), kWave - 1) << 32) |
                                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fb, kWave - 1)));
            const bool any_nan = __any(nn != 0u);
            if (lane == 0) {
                if (ct) {
                    atomicAdd(&sh.low_cnt, ct);
                    atomicAdd(&sh.low_sum, ft);
                }
                if (any_nan) atomicOr(&sh.low_nan, 1u);
            }
            ++rare_n;
        }
        rare = 0u;
    });
    if (tail != 0u) flush();
    if (rare_n != 0u && lane == 0) atomicAdd(&ctl->reserved, rare_n);   // (statistics: tiles holding values outside the window)
    // per-wave ranges (the values outside the window are in sh.low_*)
    const float wmn = wave_min(mn), wmx = wave_max(mx);
    if (lane == 0) {
        sh.red_mn[w] = wmn;
        sh.red_mx[w] = wmx;
    }
}

#endif   // DPL_WITH_ONEREAD
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
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {   // per-lane values below 2^56
    const uint32_t lo = wave_sum_dpp((uint32_t)v & 0xFFFFFFu), hi = wave_sum_dpp((uint32_t)(v >> 24));
    return (unsigned long long)lo + ((unsigned long long)hi << 24);
}

// The exact walk of one pair (one workgroup per pair, largest pairs first): suffix totals of the pair's merged row (the row is
// handed back zeroed), s_0, then the reference's iteration — totals of the bins above the iterate's bin (exact integers)
// + the listed values of that bin (integer mantissa sums) — verifying that every iterate lands in a gathered bin.  Records
// the bins it stepped into (or, when it left the gathered set, the pair's bracket) for the next batches.
template <int kVecT>
__device__ __forceinline__ void walk_pair(
    const uint32_t pair, double* s_ge, uint32_t* n_ge, Shared& sh,
    dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl,
    const unsigned long long* __restrict__ lh, const uint32_t* __restrict__ pair_slice0, const PredRows pred,
    uint32_t* __restrict__ vis_w, uint32_t n_tensors, const uint64_t* __restrict__ pair_base,
    const float* __restrict__ list0, const dpl_work_item* __restrict__ slices, int dynamic_sym, int max_iters, int fail_every,
    int phase, uint32_t* __restrict__ rescue_bm, uint32_t* __restrict__ missed, const float* __restrict__ list_rescue,
    const uint32_t* __restrict__ pred_t, float* __restrict__ tstat, unsigned long long* __restrict__ resc, uint32_t fused_cnt) {
    // phase 3: FUSED — called by the streaming workgroup of a single-slice pair: the histogram is in LDS already (s_ge's memory,
    // packed), the statistics in sh.f_*, the list (one segment of sh.cursor entries) was written by this workgroup.
    // phase 0: the walk of every pair.  phase 1 (only_missed): the pass behind k_octav_walk_sorted, see below.  phase 2: the
    // RESCUE walk — the pairs phase 0 / 1 could not finish (mode 3), over the values k_octav_rescue_gather collected for them
    // from a second read of those pairs alone: the bins of the pair's exact bracket (rescue_bm), one list (list_rescue).
    const bool only_missed = phase == 1, rescue = phase == 2, fused = phase == 3;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    dpl_octav_state* me = st + pair;
    // only_missed: the pass behind k_octav_walk_sorted — the pairs that kernel could not finish (mode 1) are walked again
    // here up to the bin that was not gathered, for the sake of what this kernel does THEN: publish the pair's bracket for
    // the next batches and leave the state the compaction route starts from.  No pair missed (the steady state): nothing to do.
    if (only_missed && (ctl->iters == 0u || me->mode != 1u)) return;
    if (rescue && (me->mode != 3u || me->done)) return;
    const unsigned long long n_pair = fused ? (unsigned long long)fused_cnt : me->n_elems;
    if (n_pair == 0ull) return;   // an empty pair: nothing was streamed
    const bool small = n_pair <= (unsigned long long)kSmallCap;
    const uint32_t tensor = pair % n_tensors;
    DPL_PROF_T(pt0);
    // What the walk reads from global memory is requested HERE, before the histogram is turned into suffix totals: inside the
    // streaming kernel a global load takes 1 - 2 us to come back and the walk used to wait for three of them one after the
    // other (the prediction row, the tensor's row for the statistics, the list).
    const float* lp = (rescue ? list_rescue : list0) + pair_base[pair];
    f4 v[kVecT];
    // a list of ONE segment at the start of the pair's region (a single-slice pair, a rescue list): rows straight from `len`,
    // no look-up of the segment table in LDS — the pairs with the longest lists (a flat distribution: the output of an erf is
    // uniform, every bin of its top octave holds 0.8 % of the pair, eight iterates list 6 - 10 % of it: 59 k values of
    // ViT-B/16's 605 184-element MLP tensors, 43 rows beyond the registers streamed in every iteration) spent a trip through
    // LDS per group of rows (ViT-B/16, batches of such images: - 2 ... - 8 %)
    auto load_rows1 = [&](auto& dst, auto count, uint32_t row0, uint32_t len) {
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
    uint32_t len_early = 0u;
    if (fused) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the list is this workgroup's own global stores (one CU, one L1)
        len_early = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.cursor);
        load_rows1(v, std::integral_constant<int, kVecT>{}, 0u, len_early);
    }
    uint32_t bm_early = 0u, lw_early = 0u;
    if (!rescue) {
        if (tid < (uint32_t)kLogWords) bm_early = small ? 0xFFFFFFFFu : pred.row(pair, tensor)[tid];
        if (pred_t) lw_early = pred_t[tensor * kPredRow + ((kLogNB - 1 - (int)tid * (kLogNB / kThreads)) >> 5)];
    }
    // per-bin totals = the sum of the pair's slice rows -> LDS (own bins per thread)
    {
        constexpr int kPerT = kLogNB / kThreads;
        const int hi = kLogNB - 1 - (int)tid * kPerT;
        const uint32_t sl0 = fused ? 0u : pair_slice0[2 * pair], sl1 = fused ? 0u : pair_slice0[2 * pair + 1];
        uint32_t cnt[kPerT];
        unsigned long long mant[kPerT];
#pragma unroll
        for (int qq = 0; qq < kPerT; ++qq) {
            cnt[qq] = 0u;
            mant[qq] = 0ull;
        }
        if (fused) {   // the slice's histogram, still in LDS (bit 63: the gather flag)
            const unsigned long long* lp = reinterpret_cast<const unsigned long long*>(s_ge);
#pragma unroll
            for (int qq = 0; qq < kPerT; ++qq) {
                const unsigned long long v = lp[hi - qq] & ~(1ull << 63);
                cnt[qq] = (uint32_t)(v >> kPackShift);
                mant[qq] = v & kPackMask;
            }
        }
        if (rescue) {   // the suffix totals the first walk left for this pair
            const double* rs = reinterpret_cast<const double*>(resc + (uint64_t)pair * kRescRow);
            const uint32_t* rn = reinterpret_cast<const uint32_t*>(resc + (uint64_t)pair * kRescRow + kLogNB);
#pragma unroll
            for (int qq = 0; qq < kPerT; ++qq) {
                n_ge[hi - qq] = rn[hi - qq];
                s_ge[hi - qq] = rs[hi - qq];
            }
        }
        for (uint32_t sl = sl0; sl < sl1 && !rescue; ++sl) {
            const unsigned long long* row = lh + (uint64_t)sl * kLogNB;
#pragma unroll
            for (int qq = 0; qq < kPerT; ++qq) {
                const unsigned long long v = row[hi - qq];
                cnt[qq] += (uint32_t)(v >> kPackShift);
                mant[qq] += v & kPackMask;
            }
        }
        if (!rescue) {
#pragma unroll
            for (int qq = 0; qq < kPerT; ++qq) {
                const int b = hi - qq;
                if (b == 0) cnt[qq] = 0u, mant[qq] = 0ull;   // bin 0 holds no element (its row words are the segment lengths)
                n_ge[b] = cnt[qq];
                s_ge[b] = bin_sum(mant[qq], cnt[qq], b);
            }
        }
        // the pair's gathered values: one list segment per slice, at the slice's element offset inside the pair
        if (rescue) {   // one list: what the rescue's gather pass wrote
            if (tid == 0) {
                sh.seg_off[0] = 0u;
                sh.seg_len[0] = me->len[0];
            }
        } else if (fused) {
            if (tid == 0) {
                sh.seg_off[0] = 0u;
                sh.seg_len[0] = sh.cursor;
            }
        } else if (tid < sl1 - sl0) {
            sh.seg_off[tid] = (uint32_t)(slices[sl0 + tid].offset - slices[sl0].offset);
            sh.seg_len[tid] = (uint32_t)lh[(uint64_t)(sl0 + tid) * kLogNB];
        }
        if (rescue) __syncthreads();
        else suffix_in_place(n_ge, s_ge, sh);
        // what the publication below adds to the bins the walk steps into: bins that hold next to nothing (cheap) and the sparse
        // tail (thin) — per bitmap word, by the threads that own the bins (eight consecutive bins: one word), while the walk's
        // list is on its way; a loop over all bins per word at publication time was a quarter of a small pair's walk
        if (tid < (uint32_t)kLogWords) {
            sh.cheapw[tid] = 0u;
            sh.thinw[tid] = 0u;
        }
        if (tid == 0) sh.would_list = 0u;
        __syncthreads();
        if (!rescue) {
            const uint32_t cheap_n = (uint32_t)(n_pair >> DPL_CHEAP_SHIFT), thin_n = (uint32_t)(n_pair >> DPL_THIN_SHIFT);
            // (also: what the tensor's prediction from earlier batches would have listed of this pair — the selection statistics)
            const uint32_t lw = lw_early;
            uint32_t cb = 0u, tb = 0u, wl = 0u;
            uint32_t above = hi + 1 < kLogNB ? n_ge[hi + 1] : 0u;
#pragma unroll
            for (int qq = 0; qq < kPerT; ++qq) {
                const int b = hi - qq;
                const uint32_t here = n_ge[b];
                const bool valid = b > 0 && b < kLogNB - 1;
                cb |= (valid && here - above <= cheap_n ? 1u : 0u) << (b & 31);
                tb |= (valid && here != 0u && here <= thin_n ? 1u : 0u) << (b & 31);
                wl += (valid && ((lw >> (b & 31)) & 1u)) ? here - above : 0u;
                above = here;
            }
            atomicOr(&sh.cheapw[hi >> 5], cb);
            atomicOr(&sh.thinw[hi >> 5], tb);
            if (wl) atomicAdd(&sh.would_list, wl);
        }
    }
    // the bins whose values were gathered (the walk may only step into these)
    if (tid < (uint32_t)kLogWords) {
        sh.bm[tid] = rescue ? rescue_bm[(uint64_t)pair * kLogWords + tid] : bm_early;
        sh.pub[tid] = 0u;
    }
    if (tid == 0 && rescue) {   // s_0 and the divisor are in the state since the first walk
        sh.s0 = me->s;
        sh.ud = me->unsigned_div;
        sh.n_elems = me->n_elems;
        // (more gathered than the pair's region of the rescue list holds: the list is incomplete — the compaction route)
        sh.route = me->len[0] > (uint32_t)(pair_base[pair + 1] - pair_base[pair]) ? 1u : 2u;
    } else if (tid == 0) {
        const double sum_out = fused ? sh.f_sum : me->sum;
        const unsigned long long nz_out = fused ? (unsigned long long)sh.f_nz : me->cnt_gt;
        const unsigned long long n = n_pair;
        const float gmn = fused ? sh.f_mn : dec_f32(me->min_enc), gmx = fused ? sh.f_mx : dec_f32(me->max_enc);
        const bool nanseen = fused ? sh.f_nan != 0u : me->nan_seen != 0u;
        // forward_net.py:319 — np.abs(data_min - 0) < 1e-6 (float32 compare) and 'dynamic_sym' in qi_params
        const float ud = (dynamic_sym && fabsf(gmn) < 1e-6f && !nanseen) ? 4.0f : 1.0f;
        // forward_net.py:324 — sum(|x|) / count(|x| > 0): exact window totals + the out-of-window part
        const float s0 = nanseen ? __uint_as_float(0x7FC00000u)
            : __fdiv_rn((float)(sum_out + s_ge[1]), (float)(long long)(nz_out + n_ge[1]));
        uint32_t route = 2u;                                         // 2: walk
        if (s0 != s0 || max_iters <= 0) route = 0u;                  // 0: finished (NaN is a fixed point)
        else if (!(fmaxf(fabsf(gmn), fabsf(gmx)) < log_edge(kLogNB))) route = 1u;   // 1: values >= 2^14 / inf: compaction route
        sh.s0 = s0;
        sh.ud = ud;
        sh.n_elems = n;
        sh.route = route;
        me->s = s0;
        me->unsigned_div = ud;
        me->iters = 0u;
        me->sum = 0.0;
        me->cnt_gt = 0ull;
        me->cnt_le = 0ull;
        me->len[0] = 0u;
        me->len[1] = 0u;
        me->cur = 2u;
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
        // the list (bit patterns of |x|) goes into registers, 1024 values per ROW (one 16-byte vector per thread); every
        // segment starts a new row; a list of more rows than the registers hold is re-read in pieces every iteration
        const uint32_t n_seg = (rescue || fused) ? 1u : __builtin_amdgcn_readfirstlane(pair_slice0[2 * pair + 1] - pair_slice0[2 * pair]);
        uint32_t n_rows = fused ? (len_early + 1023u) >> 10 : 0u, L = fused ? len_early : 0u;
        for (uint32_t j = 0; j < n_seg && !fused; ++j) {
            const uint32_t len = __builtin_amdgcn_readfirstlane(sh.seg_len[j]);
            n_rows += (len + 1023u) >> 10;
            L += len;
        }
        // the first kVec rows stay in registers for the whole walk; the rows beyond (the 3 % lists of the largest pairs) are
        // streamed kOver at a time in every iteration — requested before the resident rows are scanned, consumed after
        f4 ov[kOver];
        auto load_rows_segs = [&](auto& dst, auto count, uint32_t row0) {
            constexpr int kN = decltype(count)::value;
            uint32_t j = 0u, r = row0;   // segment and row inside it of row `row0`
            while (j < n_seg) {
                const uint32_t rows_j = (__builtin_amdgcn_readfirstlane(sh.seg_len[j]) + 1023u) >> 10;
                if (r < rows_j) break;
                r -= rows_j;
                ++j;
            }
            const uint32_t voff = tid << 4;
#pragma unroll
            for (int u = 0; u < kN; ++u) {
                uint32_t len = j < n_seg ? __builtin_amdgcn_readfirstlane(sh.seg_len[j]) : 0u;
                while (j < n_seg && (r << 10) >= len) {   // past the segment's end (or an empty segment): the next one
                    ++j;
                    r = 0u;
                    len = j < n_seg ? __builtin_amdgcn_readfirstlane(sh.seg_len[j]) : 0u;
                }
                // buffer loads: zero fill past the segment's end (one descriptor per row: the range check leaves the SGPR
                // offset out, so the row offset goes into the base)
                const float* p = lp;
                int nbytes = 0;
                if (j < n_seg) {
                    p = lp + __builtin_amdgcn_readfirstlane(sh.seg_off[j]) + (r << 10);
                    nbytes = (int)(min(len - (r << 10), 1024u) << 2);
                    ++r;
                }
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
                dst[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
            }
        };
        auto load_rows = [&](auto& dst, auto count, uint32_t row0) {
            if (n_seg == 1u) load_rows1(dst, count, row0, L);   // (uniform)
            else load_rows_segs(dst, count, row0);
        };
        if (!fused) load_rows(v, std::integral_constant<int, kVecT>{}, 0u);   // (fused: requested at the top)
        // Every wave takes the step itself from the four partial sums (one barrier and two LDS round trips per iteration instead
        // of two and five): inside the streaming kernel the LDS pipeline is full of the other workgroups' histogram atomics and
        // a round trip costs several hundred cycles.  The gathered-bin bitmap sits in registers (lane l: word l).
        const uint32_t bm_reg = sh.bm[lane];
        auto marked = [&](int j) {
            return j > 0 && j < kLogNB - 1 && (((uint32_t)__builtin_amdgcn_readlane((int)bm_reg, j >> 5) >> (j & 31)) & 1u);
        };
        // all waves count their share of the list and leave their partial sums in LDS; which wave records the bins entered: by
        // workgroup
        const int stepper = (int)(blockIdx.x & (kWaves - 1));
        int jb = log_bin(s);
        bad = marked(jb) ? 0u : 1u;
        if (fail_every > 0 && pair % (uint32_t)fail_every == 0u) bad = 1u;   // test hook: the restart path
        unsigned long long n_above = 0ull;
        double s_above = 0.0;
        auto enter = [&](int j) {   // exact totals of the bins above bin j; bin j goes on record
            n_above = (j + 1 < kLogNB) ? (unsigned long long)n_ge[j + 1] : 0ull;
            s_above = (j + 1 < kLogNB) ? s_ge[j + 1] : 0.0;
            if (lane == 0 && w == stepper) sh.pub[j >> 5] |= 1u << (j & 31);   // (every wave enters; one records)
        };
        if (!bad) enter(jb);
        uint32_t par = 0u;   // alternating slots: a wave may write iteration k + 1's partials while another still reads k's
        uint32_t done = 0u;
        DPL_PROF_T(pt2);
        DPL_PROF_ADD(1, pt1, pt2);
        if (tid == 0) {
            g_prof_iters_add(blockIdx.x, 0u);
            DPL_PROF_L(L);
            if (phase == 0 || fused) {
                atomicAdd(&ctl->sum, (double)L);   // the batch's gathered values: what the caller's form choice looks at
                if (!small && tstat && pred.use && pred.use[tensor]) {   // ... and what the pair's own sample made it list
                    atomicAdd(tstat + (size_t)tensor * kTstatRow + 5, (float)L);
                    atomicAdd(tstat + (size_t)tensor * kTstatRow + 6, (float)n_elems);
                    atomicAdd(tstat + (size_t)tensor * kTstatRow + 10, 1.0f);
                }
            }
        }
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
            if (n_rows > (uint32_t)kVecT) load_rows(ov, std::integral_constant<int, kOver>{}, (uint32_t)kVecT);
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
                if (r0 + (uint32_t)kOver < n_rows) load_rows(ov, std::integral_constant<int, kOver>{}, r0 + (uint32_t)kOver);
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
    DPL_PROF_T(pt4);
    // ---- what the next batches should gather for this tensor: the bins this walk stepped into — or, when it
    // left the gathered set, the pair's bracket over the bin edges (histogram only) — plus neighbours that hold
    // next to nothing.
    // A pair that left the gathered set is RESCUED: its exact bracket (+ the same cheap extras) goes to its rescue row and the
    // pair on the list of k_octav_rescue_gather, which re-reads this pair alone; only a bracket that cannot be formed (a flat
    // distribution, values beyond the window) or a rescue walk that fails sends the pair to the compaction route.
    bool rescued = false;
    if (!small && route == 2u && !rescue) {
        if (bad) {
            if (tid < (uint32_t)kLogWords) sh.pub[tid] = 0u;
            __syncthreads();
            if (tid == 0) sh.route = bracket_marks(n_ge, s_ge, sh.pub, sh.s0, sh.ud, sh.n_elems).route;
            __syncthreads();
            rescued = sh.route == 2u;
        }
        if (tid < (uint32_t)kLogWords) {
            // neighbours: bin j-1 / j+1 of a published bin j join when they hold <= 0.05 % of the pair (sh.cheapw)
            const uint32_t mine = sh.pub[tid];
            const uint32_t up = (mine << 1) | (tid > 0 ? sh.pub[tid - 1] >> 31 : 0u);                    // j + 1 candidates
            const uint32_t dn = (mine >> 1) | (tid + 1 < (uint32_t)kLogWords ? sh.pub[tid + 1] << 31 : 0u);   // j - 1 candidates
            const uint32_t add = (up | dn) & ~mine & sh.cheapw[tid];
            // the sparse tail, wholesale: every bin from which on no more than 1/512 of the pair lies above — that is where
            // the late iterates land, and where they scatter most from image to image (sh.thinw)
            const uint32_t tail = sh.thinw[tid];
            const uint32_t out = mine | add | tail;
            if (out) atomicOr(vis_w + tensor * kLogWords + tid, out);
            if (rescued) rescue_bm[(uint64_t)pair * kLogWords + tid] = out;
            // (these 64 threads are wave 0)  What the tensor's prediction FROM EARLIER BATCHES would have listed for this pair and
            // whether it would have covered the bins this walk needed: k_octav_oneread_init chooses per tensor between that
            // prediction and the one from a sample of the pair itself (k_octav_probe)
            const uint32_t lw = pred_t[tensor * kPredRow + tid];
            const uint32_t would_list = sh.would_list;
            const bool would_miss = __builtin_amdgcn_ballot_w64((mine & ~lw) != 0u) != 0ull;
            if (tid == 0) {
                float* ts = tstat + (size_t)tensor * kTstatRow;
                atomicAdd(ts + 0, (float)would_list);
                atomicAdd(ts + 1, (float)sh.n_elems);
                atomicAdd(ts + 2, would_miss ? 1.0f : 0.0f);
                atomicAdd(ts + 3, 1.0f);
            }
        }
    }
    if (rescued) {   // the rescue walk starts from these suffix totals (it has no histogram of its own)
        double* rs = reinterpret_cast<double*>(resc + (uint64_t)pair * kRescRow);
        uint32_t* rn = reinterpret_cast<uint32_t*>(resc + (uint64_t)pair * kRescRow + kLogNB);
        for (int b = tid; b < kLogNB; b += kThreads) {
            rs[b] = s_ge[b];
            rn[b] = n_ge[b];
        }
    }
    DPL_PROF_T(pt5);
    DPL_PROF_ADD(3, pt4, pt5);
    if (tid == 0) {
        if (route == 0u) {
            me->done = 1u;
            me->mode = 2u;
        } else if (bad && rescued) {
            if (!rescue && tstat && pred.use && pred.use[tensor]) atomicAdd(tstat + (size_t)tensor * kTstatRow + 9, 1.0f);
            // restart from s_0 (in me->s) on the pair's exact bracket: its units go on the rescue's work list
            me->mode = 3u;
            me->done = 0u;
            me->len[0] = 0u;
            const uint32_t nu = (uint32_t)((sh.n_elems + kRescueUnit - 1) / kRescueUnit);
            const uint32_t e = atomicAdd(&ctl->len[0], 1u), u0 = atomicAdd(&ctl->len[1], nu);
            missed[3 * e] = pair;
            missed[3 * e + 1] = u0;
            missed[3 * e + 2] = nu;
        } else if (bad) {
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

#ifdef DPL_WITH_ONEREAD   // the round-3 form's streaming kernel and first walk
// K1: one workgroup per slice (largest pairs first).  A plain grid rather than a persistent loop: the hardware scheduler is
// then free to interleave workgroups of the previous batch's walk kernel (second stream) with these.
// A slice that is a WHOLE pair (all but the largest tensors) is walked right here (fuse != 0): its histogram is still in LDS
// (converted in place into the suffix totals), its list still in L2 — no histogram row goes out and comes back, no second
// launch has to find a slot beside the next batch's streaming kernel.
struct FusedArgs {
    uint32_t* vis_w;
    uint32_t* rescue_bm;
    uint32_t* missed;
    float* tstat;
    unsigned long long* resc;
    int dynamic_sym, max_iters, fail_every, fuse;
};
__global__ __launch_bounds__(kThreads, DPL_RES_OCC) void k_octav_oneread(
    const dpl_work_item* __restrict__ slices, const float* const* __restrict__ segs, dpl_octav_state* __restrict__ st,
    unsigned long long* __restrict__ lh, const PredRows pred, uint32_t n_tensors,
    const uint64_t* __restrict__ pair_base, const uint32_t* __restrict__ pair_slice0, float* __restrict__ list0,
    dpl_octav_state* __restrict__ ctl, const FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* l_packed = reinterpret_cast<unsigned long long*>(lds_raw);
    __shared__ Shared sh;

    const uint32_t tid = threadIdx.x;
    DPL_PROF_T(kt0);
    const dpl_work_item it = slices[blockIdx.x];
    const uint32_t pair = it.slot, n_sl = it.reserved, cnt = it.count;
    dpl_octav_state* me = st + pair;
    const float* pg = segs[it.seg] + it.offset;
    const bool small = n_sl == 1u && cnt <= kSmallCap;    // the walk holds the pair's whole window in registers: no prediction
    const uint32_t tensor = pair % n_tensors;
    // empty histogram; bit 63 of a bin's word = "gather this bin's values": the pair's prediction row
    // (a small pair gathers its whole window)
    const uint32_t* prow = pred.row(pair, tensor);
    for (int b = tid; b < kLogNB; b += kThreads) {
        const uint32_t f = small ? 1u : (prow[b >> 5] >> (b & 31)) & 1u;
        l_packed[b] = (unsigned long long)f << 63;
    }
    if (tid < (uint32_t)kWave) l_packed[kLogNB + tid] = 0ull;
    if (tid == 0) {
        sh.cursor = 0u;
        sh.low_sum = 0.0;
        sh.low_cnt = 0u;
        sh.low_nan = 0u;
    }
    __syncthreads();

    // ------------------------------------------------------------------ 1. the slice's only HBM read, tile by tile
    // the slice's region of the pair's list: at the slice's element offset inside the pair
    const uint64_t in_pair = it.offset - slices[pair_slice0[2 * pair]].offset;
    stream_slice(pg, cnt, reinterpret_cast<uint32_t*>(list0 + pair_base[pair] + in_pair), sh, ctl);
    __syncthreads();   // every LDS histogram atomic of the slice has landed; the per-wave statistics are in sh
    DPL_PROF_T(kt1);
    DPL_PROF_ADD(4, kt0, kt1);

    // ------------------------------------------------------------------ 2. publish the slice (nothing waits for it)
    if (tid == 0) {
        float tmn = INFINITY, tmx = -INFINITY;
        const uint32_t tnz = sh.low_cnt, tnan = sh.low_nan;
        const double tsum = sh.low_sum;
        for (int j = 0; j < kWaves; ++j) {
            tmn = fminf(tmn, sh.red_mn[j]);
            tmx = fmaxf(tmx, sh.red_mx[j]);
        }
        if (tnz) {
            atomicAdd(&me->sum, tsum);
            atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)tnz);
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&me->n_elems), (unsigned long long)cnt);
        if (tmn <= tmx) {
            atomicMin(&me->min_enc, enc_f32(tmn));
            atomicMax(&me->max_enc, enc_f32(tmx));
        }
        if (tnan) atomicOr(&me->nan_seen, 1u);
        sh.f_sum = tsum;
        sh.f_nz = tnz;
        sh.f_nan = tnan;
        sh.f_mn = tmn;
        sh.f_mx = tmx;
    }
    if (fa.fuse && n_sl == 1u) {
        // ------------------------------------------------------------------ 3. the pair's walk, here and now
        // s_ge takes the histogram's own 16 KiB (a thread reads its eight packed words before it writes its eight doubles over
        // them), n_ge the queues' space (all flushed)
        __syncthreads();
        walk_pair<kVec>(pair, reinterpret_cast<double*>(lds_raw), reinterpret_cast<uint32_t*>(lds_raw + kLdsA), sh, st, ctl, nullptr, pair_slice0,
                  pred, fa.vis_w, n_tensors, pair_base, list0, slices, fa.dynamic_sym, fa.max_iters, fa.fail_every, 3, fa.rescue_bm,
                  fa.missed, nullptr, pred.t, fa.tstat, fa.resc, cnt);
        DPL_PROF_T(kt2);
        DPL_PROF_ADD(5, kt1, kt2);
        return;
    }
    // the slice's histogram goes out as ONE row of plain, coalesced stores (16 KiB, empty bins included: nothing to zero
    // beforehand, no read-modify-write at the memory side); the walk adds up the rows of a pair's slices
    unsigned long long* row = lh + (uint64_t)blockIdx.x * kLogNB;
    // (bin 0 holds no element: its word carries the length of the slice's list segment)
    for (int b = tid; b < kLogNB; b += kThreads) row[b] = b == 0 ? (unsigned long long)sh.cursor : l_packed[b] & ~(1ull << 63);
}

// kVecT = 16: any pair (four workgroups per CU).  kVecT = 32 (two per CU): the multi-slice pairs the fused schedule leaves to this
// kernel — their lists (2.8 % of 802 816 elements: 22 k values) then sit in registers whole instead of being streamed from L2 in
// every iteration beyond the first 16 Ki values.
template <int kVecT>
__global__ __launch_bounds__(kThreads, kVecT > 16 ? 2 : DPL_WALK_OCC) void k_octav_walk(
    dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl, const uint32_t* __restrict__ pair_order,
    const unsigned long long* __restrict__ lh, const uint32_t* __restrict__ pair_slice0, const PredRows pred,
    uint32_t* __restrict__ vis_w, uint32_t n_tensors, const uint64_t* __restrict__ pair_base,
    const float* __restrict__ list0, const dpl_work_item* __restrict__ slices, int dynamic_sym, int max_iters, int fail_every,
    int phase, uint32_t* __restrict__ rescue_bm, uint32_t* __restrict__ missed, const uint32_t* __restrict__ pred_t,
    float* __restrict__ tstat, unsigned long long* __restrict__ resc) {
    __shared__ double s_ge[kLogNB];
    __shared__ uint32_t n_ge[kLogNB];
    __shared__ Shared sh;
    walk_pair<kVecT>(pair_order ? pair_order[blockIdx.x] : blockIdx.x, s_ge, n_ge, sh, st, ctl, lh, pair_slice0, pred, vis_w, n_tensors,
              pair_base, list0, slices, dynamic_sym, max_iters, fail_every, phase, rescue_bm, missed, nullptr, pred_t, tstat, resc, 0u);
}

#endif   // DPL_WITH_ONEREAD
// The rescue walk (phase 2): a small persistent grid over the list of rescued pairs — usually empty, and a launch that has
// nothing to do should not have thousands of workgroups to schedule between those of the next batch's streaming kernel.
__global__ __launch_bounds__(kThreads, DPL_WALK_OCC) void k_octav_walk_rescue(
    dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl, const unsigned long long* __restrict__ lh,
    const uint32_t* __restrict__ pair_slice0, uint32_t n_tensors, const uint64_t* __restrict__ pair_base,
    const dpl_work_item* __restrict__ slices, int dynamic_sym, int max_iters, int fail_every, uint32_t* __restrict__ rescue_bm,
    uint32_t* __restrict__ missed, const float* __restrict__ list_rescue, unsigned long long* __restrict__ resc) {
    __shared__ double s_ge[kLogNB];
    __shared__ uint32_t n_ge[kLogNB];
    __shared__ Shared sh;
    const uint32_t n_missed = ctl->len[0];
    for (uint32_t e = blockIdx.x; e < n_missed; e += gridDim.x) {
        walk_pair<kVec>(missed[3 * e], s_ge, n_ge, sh, st, ctl, lh, pair_slice0, PredRows{nullptr, nullptr, nullptr}, nullptr, n_tensors, pair_base, nullptr, slices,
                  dynamic_sym, max_iters, fail_every, 2, rescue_bm, missed, list_rescue, nullptr, nullptr, resc, 0u);
        __syncthreads();
    }
}


#ifdef DPL_WITH_ONEREAD   // the round-3 form's sorted-run walk, state initialisation and sample-based prediction
// ---------------------------------------------------------------------------------------------------------------------
// Sorted runs.  k_octav_sort: one workgroup per slice; the slice's list (as gathered: arrival order) is sorted IN PLACE,
// kChunk values at a time, by the rank of the values' bins — a counting sort staged in LDS: the rank's counter hands out
// the position inside the rank (returning LDS add), an exclusive scan of the counters the rank's start, the values are
// placed in the LDS stage and leave as full 16-byte stores; the starts go to the chunk's directory row.  Random placement
// happens in LDS only (64 lanes to 64 global lines would cost 64 cycles per store instruction).
// k_octav_walk_sorted then needs, per iteration, the directory entries of ONE rank and the few hundred values behind them —
// one WAVE walks a pair, no list in registers, no barrier, and a wide prediction costs bandwidth here instead of scan time.
__global__ __launch_bounds__(kThreads) void k_octav_sort(
    const dpl_work_item* __restrict__ slices, const uint32_t* __restrict__ pair_slice0, const unsigned long long* __restrict__ lh,
    const PredRows pred, uint32_t n_tensors, const uint64_t* __restrict__ pair_base, float* __restrict__ list0,
    const uint32_t* __restrict__ slice_chunk0, uint16_t* __restrict__ dir, int fuse) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[kChunk];
    __shared__ uint32_t cnt[kMaxFlag], off[kMaxFlag + 1];
    __shared__ unsigned long long bp[kLogWords];   // per word: bitmap (low half) | ranks below the word (high half)
    const uint32_t tid = threadIdx.x;
    const dpl_work_item it = slices[blockIdx.x];
    if (it.reserved == 1u && (fuse || it.count <= kSmallCap)) return;   // walked by its streaming workgroup / a small pair (registers)
    const uint32_t len = (uint32_t)lh[(uint64_t)blockIdx.x * kLogNB];
    if (len == 0u) return;
    const uint32_t pair = it.slot, tensor = pair % n_tensors;
    uint32_t* region = reinterpret_cast<uint32_t*>(list0 + pair_base[pair] + (it.offset - slices[pair_slice0[2 * pair]].offset));
    if (tid < (uint32_t)kLogWords)
        bp[tid] = (unsigned long long)pred.row(pair, tensor)[tid] | ((unsigned long long)pred.row(pair, tensor)[kLogWords + tid] << 32);
    constexpr int kPer = (int)(kChunk / kThreads / 4);   // 16-byte vectors per thread and chunk
    for (uint32_t c0 = 0; c0 < len; c0 += kChunk) {
        const uint32_t n = min(len - c0, kChunk);
        if (tid < (uint32_t)kMaxFlag) cnt[tid] = 0u;
        // the chunk -> registers (zero fill past its end: a zero is no entry)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(region + c0), 0, (int)(n << 2), 0x00020000);
        uint32_t val[kPer * 4], rp[kPer * 4];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const f4 x = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid + (uint32_t)k * kThreads) << 4, 0, 0));
            val[4 * k + 0] = __float_as_uint(x.x);
            val[4 * k + 1] = __float_as_uint(x.y);
            val[4 * k + 2] = __float_as_uint(x.z);
            val[4 * k + 3] = __float_as_uint(x.w);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer * 4; ++e) {   // rank of the value's bin, position inside the rank
            const uint32_t u = val[e];
            rp[e] = 0xFFFFFFFFu;
            if (u != 0u) {
                const uint32_t b = (u >> kLogShift) - kLogKey0;
                const unsigned long long wv = bp[(b >> 5) & (uint32_t)(kLogWords - 1)];
                const uint32_t r = (uint32_t)(wv >> 32) + (uint32_t)__popc((uint32_t)wv & ((1u << (b & 31u)) - 1u));
                rp[e] = (min(r, (uint32_t)(kMaxFlag - 1)) << 16) | atomicAdd(&cnt[min(r, (uint32_t)(kMaxFlag - 1))], 1u);
            }
        }
        __syncthreads();
        if (tid < (uint32_t)kWave) {   // exclusive scan of the counters: four per lane
            const uint32_t a0 = cnt[4 * tid], a1 = cnt[4 * tid + 1], a2 = cnt[4 * tid + 2], a3 = cnt[4 * tid + 3];
            const uint32_t incl = wave_incl_scan_dpp(a0 + a1 + a2 + a3), base = incl - (a0 + a1 + a2 + a3);
            off[4 * tid] = base;
            off[4 * tid + 1] = base + a0;
            off[4 * tid + 2] = base + a0 + a1;
            off[4 * tid + 3] = base + a0 + a1 + a2;
            if (tid == kWave - 1) off[kMaxFlag] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer * 4; ++e)
            if (rp[e] != 0xFFFFFFFFu) stage[off[rp[e] >> 16] + (rp[e] & 0xFFFFu)] = val[e];
        __syncthreads();
        // the sorted chunk back over itself (every thread has long read its part), the starts into the directory
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t i = (tid + (uint32_t)k * kThreads) * 4u;
            if (i + 3u < n) {
                *reinterpret_cast<uint4*>(region + c0 + i) = *reinterpret_cast<const uint4*>(stage + i);
            } else {
                for (uint32_t q = i; q < n; ++q) region[c0 + q] = stage[q];
            }
        }
        uint16_t* drow = dir + (uint64_t)(slice_chunk0[blockIdx.x] + c0 / kChunk) * kDirRow;
        for (uint32_t i = tid; i <= (uint32_t)kMaxFlag; i += kThreads) drow[i] = (uint16_t)off[i];
        __syncthreads();
    }
}

// wave64 inclusive prefix sum of doubles by DPP (the two halves moved separately; lanes a step does not reach add +0.0)
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
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

constexpr int kMaxRuns = kWave;   // sorted chunks of a pair one wave handles (one per lane); more: the compaction route

// The exact walk of one pair by ONE WAVE over the pair's sorted runs (all pairs but the small ones, largest first).
//   1  the pair's slice rows -> per-bin totals -> totals ABOVE every gathered bin, by rank, in LDS (lane l owns 16 bins per
//      half of the window, highest bins in lane 0: two passes, a DPP prefix scan over the lanes each); the bins that hold
//      next to nothing and the sparse tail, for what is published afterwards;
//   2  s_0, then the reference's iteration: per step the directory entries of the iterate's rank (lane = run), a flat
//      index over the runs' segments, up to four values per lane and round in flight, count / offset sum as in k_octav_walk;
//      every iterate is verified to lie in a gathered bin;
//   3  the bins stepped into (+ neighbours holding next to nothing, + the sparse tail) are published for the next batches.
// A pair that cannot finish here (a bin that was not gathered, values beyond the window, more than kMaxRuns runs) is only
// MARKED (mode 1, counted in the control block): k_octav_walk(only_missed) publishes its bracket and prepares its state.
__global__ __launch_bounds__(kWave, DPL_SORTED_OCC) void k_octav_walk_sorted(
    dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl, const uint32_t* __restrict__ pair_order,
    const unsigned long long* __restrict__ lh, const uint32_t* __restrict__ pair_slice0, const PredRows pred,
    uint32_t* __restrict__ vis_w, uint32_t n_tensors, const uint64_t* __restrict__ pair_base, const float* __restrict__ list0,
    const dpl_work_item* __restrict__ slices, const uint32_t* __restrict__ slice_chunk0, const uint16_t* __restrict__ dir,
    int dynamic_sym, int max_iters, int fail_every, const uint32_t* __restrict__ pred_t, float* __restrict__ tstat) {
    __shared__ double t_s[kMaxFlag + 1];
    __shared__ uint32_t t_n[kMaxFlag + 1];
    __shared__ uint32_t bm[kLogWords], pre[kLogWords], cheapw[kLogWords], thinw[kLogWords], pub[kLogWords];
    __shared__ double s1_sum;
    __shared__ uint32_t s1_cnt;
    constexpr int kLdsRuns = 4;   // directory rows kept in LDS (requested before phase 1, there when the walk starts)
    __shared__ __attribute__((aligned(16))) uint16_t dirl[kLdsRuns][kDirRow];
    const uint32_t lane = threadIdx.x;
    const uint32_t pair = pair_order ? pair_order[blockIdx.x] : blockIdx.x;
    dpl_octav_state* me = st + pair;
    const unsigned long long n_pair = me->n_elems;
    if (n_pair == 0ull) return;   // an empty pair: nothing was streamed
    const uint32_t tensor = pair % n_tensors;
    const uint32_t sl0 = pair_slice0[2 * pair], sl1 = pair_slice0[2 * pair + 1];
    bm[lane] = pred.row(pair, tensor)[lane];
    pre[lane] = pred.row(pair, tensor)[kLogWords + lane];
    cheapw[lane] = 0u;
    thinw[lane] = 0u;
    pub[lane] = 0u;
    // the pair's runs: lane r holds run r (its first value relative to the pair's list, its directory row)
    uint32_t my_addr = 0u, my_dir = 0u, n_runs = 0u, listed = 0u;
    for (uint32_t sl = sl0; sl < sl1; ++sl) {
        const uint32_t len = __builtin_amdgcn_readfirstlane((uint32_t)lh[(uint64_t)sl * kLogNB]);
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(slices[sl].offset - slices[sl0].offset));
        const uint32_t ch0 = __builtin_amdgcn_readfirstlane(slice_chunk0[sl]);
        listed += len;
        for (uint32_t c = 0; c * kChunk < len; ++c) {
            if (lane == n_runs) {
                my_addr = off + c * kChunk;
                my_dir = ch0 + c;
            }
            ++n_runs;
        }
    }
    if (lane == 0) atomicAdd(&ctl->sum, (double)listed);   // the batch's gathered values: what the caller's form choice looks at
    {   // the first runs' directory rows -> LDS (16 bytes per lane and run: kDirRow / 8 lanes)
        uint4 dv[kLdsRuns];
#pragma unroll
        for (int j = 0; j < kLdsRuns; ++j) {
            const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)my_dir, j);
            dv[j] = make_uint4(0u, 0u, 0u, 0u);
            if ((uint32_t)j < n_runs && lane < (uint32_t)(kDirRow / 8))
                dv[j] = reinterpret_cast<const uint4*>(dir + (uint64_t)dj * kDirRow)[lane];
        }
#pragma unroll
        for (int j = 0; j < kLdsRuns; ++j)
            if (lane < (uint32_t)(kDirRow / 8)) reinterpret_cast<uint4*>(dirl[j])[lane] = dv[j];
    }
    __syncthreads();
    DPL_PROF_T(qt0);
    // ---- 1. totals above every gathered bin
    const uint32_t cheap_n = (uint32_t)(n_pair >> DPL_CHEAP_SHIFT), thin_n = (uint32_t)(n_pair >> DPL_THIN_SHIFT);
    uint32_t carry_n = 0u, would_list = 0u;
    double carry_s = 0.0;
    // All 16 bins of a lane — a quarter octave — share the exponent, so sums stay INTEGERS (explicit mantissas + count * 2^23,
    // below 2^51 per octave) inside an octave: a conversion to fp64 happens once per octave (its total) and once per
    // gathered bin (the integer part above it inside its octave), not once per bin (fp64 conversions run at a quarter of the
    // rate and were most of this phase: 41 us per pair with one per bin and sweep).
    for (int half = 1; half >= 0; --half) {
        const int hb = half * (kLogNB / 2) + (int)(kWave - 1 - lane) * 16;   // the lane's 16 bins; lane 0: the highest
        uint32_t cnt[16];
        unsigned long long m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            cnt[q] = 0u;
            m[q] = 0ull;
        }
        for (uint32_t sl = sl0; sl < sl1; ++sl) {
            const unsigned long long* row = lh + (uint64_t)sl * kLogNB + hb;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned long long v = row[q];
                cnt[q] += (uint32_t)(v >> kPackShift);
                m[q] += v & kPackMask;
            }
        }
        if (hb == 0) {   // bin 0 holds no element (its row words are the segment lengths)
            cnt[0] = 0u;
            m[0] = 0ull;
        }
        uint32_t tot_n = 0u;
        unsigned long long tot_m = 0ull;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            m[q] += (unsigned long long)cnt[q] << 23;   // full 24-bit mantissas
            tot_n += cnt[q];
            tot_m += m[q];
        }
        const uint32_t incl_n = wave_incl_scan_dpp(tot_n);
        // inside the octave (four lanes): the integer total of the lanes above, and the octave's total
        const uint32_t pos = lane & 3u;
        const unsigned long long p1 = __shfl_up(tot_m, 1, 4), p2 = __shfl_up(tot_m, 2, 4), p3 = __shfl_up(tot_m, 3, 4);
        const unsigned long long above_m = (pos >= 1u ? p1 : 0ull) + (pos >= 2u ? p2 : 0ull) + (pos >= 3u ? p3 : 0ull);
        unsigned long long oct_m = tot_m + __shfl_xor(tot_m, 1, 4);
        oct_m += __shfl_xor(oct_m, 2, 4);
        const double scale = log_bin_scale(hb);
        const double incl_s = wave_incl_scan_f64(pos == 3u ? (double)oct_m * scale : 0.0);   // octave totals, at the octaves' last lanes
        const double excl_s = __shfl_up(incl_s, 1, kWave);
        const double base_s = carry_s + (lane == 0 ? 0.0 : excl_s);   // everything in the octaves above the lane's
        uint32_t run_n = carry_n + incl_n - tot_n;                    // everything in the bins above the lane's
        unsigned long long run_m = above_m;
        const uint32_t word = (uint32_t)hb >> 5, sh16 = (uint32_t)hb & 16u;
        const uint32_t bits = (bm[word] >> sh16) & 0xFFFFu;
        const uint32_t below = pre[word] + (uint32_t)__popc(bm[word] & ((1u << sh16) - 1u));
        uint32_t cheap_bits = 0u, thin_bits = 0u;
        const uint32_t lbits = (pred_t[tensor * kPredRow + word] >> sh16) & 0xFFFFu;   // the tensor's prediction from earlier batches
#pragma unroll
        for (int q = 15; q >= 0; --q) {
            would_list += ((lbits >> q) & 1u) ? cnt[q] : 0u;
            if ((bits >> q) & 1u) {   // a gathered bin: totals above it, by rank
                const uint32_t r = below + (uint32_t)__popc(bits & ((1u << q) - 1u));
                t_n[r] = run_n;
                t_s[r] = base_s + (double)run_m * scale;
            }
            run_n += cnt[q];
            run_m += m[q];
            cheap_bits |= (cnt[q] <= cheap_n ? 1u : 0u) << q;
            thin_bits |= (run_n != 0u && run_n <= thin_n ? 1u : 0u) << q;
            if (hb + q == 1) {   // forward_net.py:324 — the window's part of sum(|x|) and count(|x| > 0)
                s1_sum = base_s + (double)run_m * scale;
                s1_cnt = run_n;
            }
        }
        atomicOr(&cheapw[word], cheap_bits << sh16);
        atomicOr(&thinw[word], thin_bits << sh16);
        carry_n += (uint32_t)__builtin_amdgcn_readlane((int)incl_n, kWave - 1);
        const unsigned long long tb = (unsigned long long)__double_as_longlong(incl_s);
        carry_s += __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb >> 32), kWave - 1) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb, kWave - 1)));
    }
    __syncthreads();
    DPL_PROF_T(qt1);
    DPL_PROF_ADD(0, qt0, qt1);
    // ---- 2. s_0 and the walk (every lane carries the same state)
    const double sum_out = me->sum;
    const unsigned long long nz_out = me->cnt_gt;
    const float gmn = dec_f32(me->min_enc), gmx = dec_f32(me->max_enc);
    const bool nanseen = me->nan_seen != 0u;
    // forward_net.py:319 — np.abs(data_min - 0) < 1e-6 (float32 compare) and 'dynamic_sym' in qi_params
    const float ud = (dynamic_sym && fabsf(gmn) < 1e-6f && !nanseen) ? 4.0f : 1.0f;
    const float s0 = nanseen ? __uint_as_float(0x7FC00000u) : __fdiv_rn((float)(sum_out + s1_sum), (float)(long long)(nz_out + s1_cnt));
    uint32_t route = 2u;                                         // 2: walk
    if (s0 != s0 || max_iters <= 0) route = 0u;                  // 0: finished (NaN is a fixed point)
    else if (!(fmaxf(fabsf(gmn), fabsf(gmx)) < log_edge(kLogNB)) || n_runs > (uint32_t)kMaxRuns) route = 1u;   // 1: the compaction route
    auto gathered = [&](int j) { return j > 0 && j < kLogNB - 1 && ((bm[j >> 5] >> (j & 31)) & 1u); };
    uint32_t bad = route == 1u ? 1u : 0u;
    float s = s0;
    uint32_t iters = 0u;
    if (route == 2u) {
        const uint32_t* lp = reinterpret_cast<const uint32_t*>(list0 + pair_base[pair]);
        int jb = log_bin(s);
        bad = gathered(jb) ? 0u : 1u;
        if (fail_every > 0 && pair % (uint32_t)fail_every == 0u) bad = 1u;   // test hook: the restart path
        if (!bad && lane == 0) pub[jb >> 5] |= 1u << (jb & 31);
        uint32_t done = 0u;
        while (!done && !bad) {
            const uint32_t r = __builtin_amdgcn_readfirstlane(pre[jb >> 5] + (uint32_t)__popc(bm[jb >> 5] & ((1u << (jb & 31)) - 1u)));
            const unsigned long long n_above = t_n[r];
            const double s_above = t_s[r];
            // values of bin jb above s: bit patterns in (bits(s), lower edge of bin jb + 1): d = u - bits(s) - 1 below `span`
            const uint32_t lo1 = __float_as_uint(s) + 1u;
            const uint32_t span = (((uint32_t)(jb + 1) + kLogKey0) << kLogShift) - lo1;
            uint32_t d0 = 0u, d1 = 0u;
            if (lane < n_runs) {
                if (lane < (uint32_t)kLdsRuns) {
                    d0 = dirl[lane][r];
                    d1 = dirl[lane][r + 1u];
                } else {
                    const uint16_t* drow = dir + (uint64_t)my_dir * kDirRow;
                    d0 = drow[r];
                    d1 = drow[r + 1u];
                }
            }
            const uint32_t cj = d1 - d0, aj = my_addr + d0;
            const uint32_t incl = wave_incl_scan_dpp(cj);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
            const uint32_t nr = min(n_runs, (uint32_t)kMaxRuns);
            uint32_t c = 0u;   // (wave-uniform)
            unsigned long long dsum = 0ull;
            // one ROUND: flat indices f0 + 64 k + lane, k < kFl — all of a round's loads are in flight together (a dense bin
            // of a large pair holds thousands of values: sixteen per lane and round; the bins of the late iterates: four)
            auto round = [&](auto kfl, uint32_t f0) {
                constexpr int kFl = decltype(kfl)::value;
                uint32_t a[kFl];
#pragma unroll
                for (int k = 0; k < kFl; ++k) a[k] = 0u;
                for (uint32_t j = 0; j < nr; ++j) {   // which run holds a flat index
                    const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)j);
                    const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)cj, (int)j);
                    if (ej <= f0 || ej - nj >= f0 + (uint32_t)kFl * kWave) continue;   // (uniform) not in this round
                    const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)aj, (int)j);
#pragma unroll
                    for (int k = 0; k < kFl; ++k) {
                        const uint32_t rel = f0 + (uint32_t)k * kWave + lane - (ej - nj);
                        if (rel < nj) a[k] = bj + rel;
                    }
                }
                uint32_t u[kFl];
#pragma unroll
                for (int k = 0; k < kFl; ++k) u[k] = f0 + (uint32_t)k * kWave + lane < total ? lp[a[k]] : 0u;
                uint32_t ds = 0u;
#pragma unroll
                for (int k = 0; k < kFl; ++k) {
                    const uint32_t d = u[k] - lo1;   // (a zero wraps far beyond span)
                    const bool in = d < span;
                    c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
                    ds += in ? d : 0u;
                }
                dsum += (unsigned long long)wave_sum_dpp(ds);
            };
            if (total <= 4u * kWave) {
                if (total != 0u) round(std::integral_constant<int, 4>{}, 0u);
            } else {
                for (uint32_t f0 = 0; f0 < total; f0 += 16u * kWave) round(std::integral_constant<int, 16>{}, f0);
            }
            const unsigned long long tc = c, tm = dsum + (unsigned long long)c * (unsigned long long)(lo1 & 0x7FFFFFu);
            const unsigned long long tg = n_above + tc;
            const double ts = s_above + (double)(tm + (tc << 23)) * log_bin_scale(jb);
            const OctavStep qs = octav_step(ts, tg, n_pair - tg, ud, s, iters, max_iters);
            s = qs.s;
            iters = qs.iters;
            done = qs.done;
            if (!done) {
                const int jn = log_bin(s);
                if (!gathered(jn)) {
                    bad = 1u;   // a bin that was not gathered (or out of the binned window): the compaction route takes over
                } else if (jn != jb) {
                    jb = jn;
                    if (lane == 0) pub[jb >> 5] |= 1u << (jb & 31);
                }
            }
        }
    }
    DPL_PROF_T(qt2);
    DPL_PROF_ADD(2, qt1, qt2);
    if (lane == 0) g_prof_iters_add(blockIdx.x, iters);
    __syncthreads();
    // ---- 3. what the next batches should gather for this tensor: the bins stepped into, neighbours that hold next to
    // nothing, the sparse tail.  (A pair that missed is published by k_octav_walk(only_missed): its bracket.)
    if (route == 2u && !bad) {
        const uint32_t mine = pub[lane];
        const uint32_t up = (mine << 1) | (lane > 0 ? pub[lane - 1] >> 31 : 0u);                                  // j + 1 candidates
        const uint32_t dn = (mine >> 1) | (lane + 1 < (uint32_t)kLogWords ? pub[lane + 1] << 31 : 0u);           // j - 1 candidates
        uint32_t valid = 0xFFFFFFFFu;
        if (lane == 0) valid &= ~1u;                               // bins 1 .. kLogNB - 2
        if (lane == (uint32_t)kLogWords - 1u) valid &= ~(1u << 31);
        const uint32_t out = mine | ((((up | dn) & ~mine & cheapw[lane]) | thinw[lane]) & valid);
        if (out) atomicOr(vis_w + tensor * kLogWords + lane, out);
        // selection statistics, as in k_octav_walk (a pair that missed is counted by its second walk there)
        const uint32_t wl = wave_sum_dpp(would_list);
        const bool would_miss = __builtin_amdgcn_ballot_w64((mine & ~pred_t[tensor * kPredRow + lane]) != 0u) != 0ull;
        if (lane == 0) {
            float* ts = tstat + (size_t)tensor * kTstatRow;
            atomicAdd(ts + 0, (float)wl);
            atomicAdd(ts + 1, (float)n_pair);
            atomicAdd(ts + 2, would_miss ? 1.0f : 0.0f);
            atomicAdd(ts + 3, 1.0f);
        }
    }
    if (lane == 0) {
        if (route == 0u || !bad) {
            me->s = route == 0u ? s0 : s;
            me->unsigned_div = ud;
            me->iters = route == 0u ? 0u : iters;
            me->sum = 0.0;
            me->cnt_gt = 0ull;
            me->cnt_le = 0ull;
            me->len[0] = 0u;
            me->len[1] = 0u;
            me->cur = 2u;
            me->done = 1u;
            me->mode = 2u;
        } else {
            me->mode = 1u;   // missed: k_octav_walk(only_missed) takes it from here (the streamed statistics stay in place)
            atomicAdd(&ctl->iters, 1u);
        }
    }
}

__global__ void k_octav_oneread_init(dpl_octav_state* st, int64_t n_pairs, uint32_t* vis_w, const uint32_t* vis_o, uint32_t* pred,
                                     int64_t vis_words, int zero_w, float* tstat, uint32_t* use_probe, int predict) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < vis_words) {   // this batch gathers what the current and the previous epoch's walks stepped into (a snapshot: the
                           // walks of this batch keep adding to vis_w while they run).  One wave = one tensor, lane = word.
        const uint32_t mine = zero_w ? 0u : vis_w[i];
        if (zero_w) vis_w[i] = 0u;
        uint32_t x = mine | vis_o[i];
        // at most kMaxFlag - 1 bins (lowest first): ranks are table indices; a walk that needs a dropped bin finishes on the
        // compaction route like any other miss
        const uint32_t lane = threadIdx.x & (kWave - 1);
        uint32_t incl = wave_incl_scan_dpp((uint32_t)__popc(x));
        uint32_t below = incl - (uint32_t)__popc(x);
        if (incl > (uint32_t)(kMaxFlag - 1)) {
            const uint32_t keep = below < (uint32_t)(kMaxFlag - 1) ? (uint32_t)(kMaxFlag - 1) - below : 0u;
            while ((uint32_t)__popc(x) > keep) x &= ~(1u << (31 - __clz(x)));
            below = min(below, (uint32_t)(kMaxFlag - 1));
        }
        const int64_t t = i / kLogWords;
        pred[t * kPredRow + lane] = x;
        pred[t * kPredRow + kLogWords + lane] = below;
        if (lane == 0) {
            // Which prediction this tensor's pairs get in this batch: the one above (what earlier images' walks visited: the
            // narrowest there is when the images are alike) or the one from a sample of the pair itself (k_octav_probe: wider,
            // costs a read of 1 / kProbeRate of the pair, but does not care how the images differ).  Judged by what the
            // first would have cost in the last batches' walks; with hysteresis; no history: the sample.
            // What the sample costs is MEASURED too (the values its pairs listed while the tensor was on it): on independent
            // values ~6 % of the elements, on the spatially smooth feature maps of a convolutional network — where the 32
            // neighbours of a chunk carry little more than one of them — 20 % and more; remembered in ts[7] while the tensor is on
            // the other prediction.
            float* ts = tstat + t * kTstatRow;
            const float listed = ts[0], elems = ts[1], misses = ts[2], walks = ts[3];
            if (ts[6] > 0.0f) ts[7] = ts[5] / ts[6];
            const float own = fminf(fmaxf(ts[7] > 0.0f ? ts[7] + (ts[11] != 0.0f ? 0.04f : 0.02f) : 0.08f, 0.05f), 0.5f);   // + the sample's read
            uint32_t probe = ts[4] != 0.0f ? 1u : 0u;
            if (walks < 0.5f) {
                probe = 1u;
            } else if (probe) {
                if (listed < 0.7f * own * elems && misses < 0.04f * walks) probe = 0u;
            } else {
                if (listed > 1.0f * own * elems || misses > 0.08f * walks) probe = 1u;
            }
            if (predict == 0) probe = 0u;   // forced: earlier batches only
            if (predict == 1) probe = 1u;   // forced: the pair's own sample
            ts[0] = 0.5f * listed;
            ts[1] = 0.5f * elems;
            ts[2] = 0.5f * misses;
            ts[3] = 0.5f * walks;
            ts[4] = probe ? 1.0f : 0.0f;
            ts[5] *= 0.5f;
            ts[6] *= 0.5f;
            // The brackets' width follows the misses it produces (the variance model errs where neighbours are correlated: on
            // feature maps z = 3 gave 0.5 % misses and 19 % listed): aimed at 2 – 6 % of the walks leaving the gathered bins —
            // a miss costs a re-read of that pair, a wider bracket costs every pair
            if (ts[10] >= 8.0f) {
                float z = ts[8] > 0.0f ? ts[8] : kProbeZ;
                const float rate = ts[9] / ts[10];
                if (rate < 0.02f) z *= 0.92f;
                else if (rate > 0.06f) z *= 1.08f;
                ts[8] = fminf(fmaxf(z, 1.5f), 4.0f);
            }
            ts[9] *= 0.5f;
            ts[10] *= 0.5f;
            // twice the sample where the sample's own lists stay long (feature maps: 12 % at 1/16 -> 9 % at 1/8 for 6 % more reading)
            if (ts[11] == 0.0f && ts[7] > 0.09f) ts[11] = 1.0f;
            else if (ts[11] != 0.0f && ts[7] > 0.0f && ts[7] < 0.045f) ts[11] = 0.0f;
            use_probe[t] = probe;
        }
    }
    if (i > n_pairs) return;  // slot n_pairs is the control block
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
    z.mode = 2u;
    z.n_elems = 0ull;
    z.len[0] = 0u;
    z.len[1] = 0u;
    z.cur = 2u;
    z.reserved = 0u;
    st[i] = z;
}


// ---------------------------------------------------------------------------------------------------------------------
// k_octav_probe: the prediction of a pair FROM THE PAIR ITSELF — one 128-byte chunk of every kProbeRate (a strided sample:
// every channel and every region of the feature map contributes), binned like the full pass (count, sum per bin) plus a
// sum of squares; then the iteration is walked on the SAMPLE's histogram (linear inside a bin) carrying the sampling
// variance of every iterate along: Var(s_{k+1}) ~ Var(tail mean above s_k) / (sampled tail count) + F'(s_k)^2 Var(s_k).
// Gathered: the bins within z standard deviations (kProbeZ, then per tensor whatever keeps 2 - 6 % of the walks leaving them)
// of every sampled iterate, and — from the first iterate on whose
// lower end fewer than kProbeThin sampled values lie — everything above (the late iterates' tail holds next to nothing).
// Nothing here needs to be exact: the walk verifies every iterate against what was gathered and a pair whose iterate
// falls outside is rescued (re-read alone).  One workgroup per pair; a tensor whose pairs use the prediction from earlier
// batches (use_probe == 0) only copies that row.
__global__ __launch_bounds__(kThreads, 6) void k_octav_probe(
    const dpl_span* __restrict__ pair_spans, const float* const* __restrict__ segs, const uint32_t* __restrict__ pred_t,
    const uint32_t* __restrict__ use_probe, uint32_t* __restrict__ pred_p, uint32_t n_tensors, int dynamic_sym, int max_iters,
    float z, const uint32_t* __restrict__ pair_order, const float* __restrict__ tstat) {
    // 24 KiB of LDS per workgroup (six per CU): the packed sample histogram, overlaid after the conversion by the suffix sums
    // of the values and of their squares (fp32: nothing here has to be exact), and the suffix counts
    __shared__ __attribute__((aligned(16))) unsigned long long packed[kLogNB];
    __shared__ uint32_t n_ge[kLogNB];
    float* s_ge = reinterpret_cast<float*>(packed);
    float* q_ge = s_ge + kLogNB;
    struct ProbeShared {
        uint32_t bm[kLogWords];
        uint32_t red_a[kWaves];
        float red_mn[kWaves];
    };
    __shared__ ProbeShared sh;
    __shared__ float red_q[kWaves], red_s[kWaves];
    __shared__ uint32_t red_n[kWaves];
    __shared__ int red_top[kWaves];
    __shared__ double red_de[kWaves][7];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const int w = tid / kWave;
    const uint32_t pair = pair_order[blockIdx.x], tensor = pair % n_tensors;   // largest pairs first
    const dpl_span sp = pair_spans[pair];
    if (sp.count <= (uint64_t)kSmallCap) return;   // gathers its whole window: no prediction row is read
    if (!use_probe[tensor]) return;                // this batch, the tensor's pairs gather by its row from earlier batches
    if (tstat && tstat[(size_t)tensor * kTstatRow + 8] > 0.0f) z = tstat[(size_t)tensor * kTstatRow + 8];   // the tensor's own width
    // ... and its sampling rate: one chunk of every kProbeRate, or twice that where the chunks' neighbours are so alike that
    // the brackets stay wide (what counts there is the number of LINES touched)
    const uint32_t rate = (tstat && tstat[(size_t)tensor * kTstatRow + 11] > 0.0f) ? kProbeRate / 2u : kProbeRate;
    uint32_t* row = pred_p + (uint64_t)pair * kPredRow;
    DPL_PROF_T(qp0);
    for (int b = tid; b < kLogNB; b += kThreads) packed[b] = 0ull;
    __syncthreads();
    // ---- the sample: chunk g (32 floats = 128 bytes, one L2 line: 64-byte chunks fetched whole lines for half the use —
    // 75 us instead of 40 for the 213 MB of a ResNet-50 batch) lies in the g-th window of 32 * kProbeRate elements behind the first
    // 16-byte boundary, at a slot of the window drawn from a hash of g: a regular stride would alias with the rows of the
    // tensor (a [197, 768] activation's rows are a multiple of a 1 KiB window: a fixed slot sees the same few dozen of its 768
    // channels in every token, and such a sample's iterates miss the pair's by whole bins)
    const float* p0 = segs[sp.seg] + sp.offset;
    const uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)p0 & 15u)) & 15u) >> 2);
    const uint32_t n = (uint32_t)sp.count - head;
    gptr_f4 pv = (gptr_f4)(p0 + head);
    constexpr uint32_t kChunkLanes = 8u;                         // lanes (16 bytes each) per chunk: 128 bytes = one L2 line
    constexpr uint32_t kChunkElems = kChunkLanes * 4u;
    const uint32_t n_chunks = n / (kChunkElems * rate);          // whole windows only (the last partial one is skipped)
    float mn = INFINITY;
    uint32_t m = 0u;
    // the 16 values of a chunk are neighbours (one token, one row of a feature map): not independent draws.  The variance of
    // the sample mean is therefore taken BETWEEN chunks and compared with what independent draws would give — the design effect
    // of cluster sampling (an extrapolation from each lane's four neighbours overstated it threefold on feature maps, whose
    // correlation falls off within the chunk); every iterate's variance is scaled by it.
    float c_sum = 0.0f, c_sq = 0.0f, e_sum = 0.0f, e_sq = 0.0f;   // per lane: chunk sums (first lane of a chunk) / element sums of |x|
    uint32_t c_n = 0u;        // chunks counted
    uint32_t o_cnt = 0u;      // non-zero values outside the binned window (a softmax output: most of them) ...
    float o_sum = 0.0f;       // ... and their sum
    auto eat = [&](float x) {
        mn = fminf(mn, x);
        const uint32_t bits = __float_as_uint(x);
        const uint32_t t = ((bits >> kLogShift) & 0x3FFFu) - (kLogKey0 + 1u);
        if (t < (uint32_t)(kLogNB - 1)) {
            atomicAdd(packed + t + 1u, (1ull << kPackShift) | (unsigned long long)(bits & 0x7FFFFFu));
        } else {   // outside the binned window: still part of s_0 = sum |x| / count(|x| > 0) (forward_net.py:324)
            const float ax = fabsf(x);
            if (ax > 0.0f && ax < INFINITY) {
                o_cnt += 1u;
                o_sum += ax;
            }
        }
    };
    constexpr int kIn = 8;    // loads in flight per lane: the sample of the largest pairs (3 136 chunks) in seven round trips
    for (uint32_t g0 = tid / kChunkLanes; g0 < n_chunks; g0 += (kThreads / kChunkLanes) * kIn) {
        f4 v[kIn];
#pragma unroll
        for (int u = 0; u < kIn; ++u) {
            const uint32_t g = g0 + (uint32_t)u * (kThreads / kChunkLanes);
#ifdef DPL_PROBE_FIXSLOT
            const uint32_t slot = 0u;
#else
            const uint32_t slot = ((g * 0x9E3779B1u) >> 16) % rate;   // where in its window chunk g lies
#endif
            v[u] = g < n_chunks ? __builtin_nontemporal_load(pv + ((size_t)g * rate + slot) * kChunkLanes + (tid & (kChunkLanes - 1u)))
                                : f4{0.f, 0.f, 0.f, 0.f};
            m += g < n_chunks ? 4u : 0u;
        }
#pragma unroll
        for (int u = 0; u < kIn; ++u) {
            eat(v[u].x);
            eat(v[u].y);
            eat(v[u].z);
            eat(v[u].w);
#ifndef DPL_PROBE_NODEFF
            if (u & 1) continue;                         // (every other load: an estimate of a ratio of variances)
            const float a0 = fabsf(v[u].x), a1 = fabsf(v[u].y), a2 = fabsf(v[u].z), a3 = fabsf(v[u].w);
            float cs = (a0 + a1) + (a2 + a3);
            e_sum += cs;
            e_sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            // the chunk's sum (its 8 lanes) in every one of them: two quad permutes and a half-row mirror
            cs += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cs), 0xB1, 0xF, 0xF, true));
            cs += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cs), 0x4E, 0xF, 0xF, true));
            cs += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cs), 0x141, 0xF, 0xF, true));
            if ((tid & (kChunkLanes - 1u)) == 0u) {
                c_sum += cs;
                c_sq += cs * cs;
                c_n += g0 + (uint32_t)u * (kThreads / kChunkLanes) < n_chunks ? 1u : 0u;
            }
#endif
        }
    }
    // (wave sums by DPP: the ds_bpermute form of ten reductions was a fifth of a small pair's time here)
#define DPL_FSTEP(ctrl, rmask, bound) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, bound))
    auto fsum = [](float v) {
        DPL_FSTEP(0xB1, 0xF, true);     // quad_perm [1,0,3,2]
        DPL_FSTEP(0x4E, 0xF, true);     // quad_perm [2,3,0,1]
        DPL_FSTEP(0x141, 0xF, true);    // row_half_mirror
        DPL_FSTEP(0x140, 0xF, true);    // row_mirror: every lane holds its row's sum
        DPL_FSTEP(0x142, 0xA, false);   // row_bcast15 -> rows 1, 3
        DPL_FSTEP(0x143, 0xC, false);   // row_bcast31 -> rows 2, 3
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    };
#undef DPL_FSTEP
    m = wave_sum_dpp(m);
    mn = wave_min(mn);
    const double w_cs = fsum(c_sum), w_cq = fsum(c_sq), w_es = fsum(e_sum), w_eq = fsum(e_sq);
    const double w_os = fsum(o_sum);
    o_cnt = wave_sum_dpp(o_cnt);
    c_n = wave_sum_dpp(c_n);
    if (lane == 0) {
        red_de[w][6] = (double)c_n;
        red_de[w][0] = w_cs;
        red_de[w][1] = w_cq;
        red_de[w][2] = w_es;
        red_de[w][3] = w_eq;
        red_de[w][4] = w_os;
        red_de[w][5] = (double)o_cnt;
        sh.red_a[w] = m;
        sh.red_mn[w] = mn;
    }
    __syncthreads();
    const uint32_t m_all = sh.red_a[0] + sh.red_a[1] + sh.red_a[2] + sh.red_a[3];
    const float smn = fminf(fminf(sh.red_mn[0], sh.red_mn[1]), fminf(sh.red_mn[2], sh.red_mn[3]));
    float deff = 1.0f, out_sum = 0.0f, out_cnt = 0.0f;
    {
        double t[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < kWaves; ++q)
            for (int i = 0; i < 7; ++i) t[i] += red_de[q][i];
        out_sum = (float)t[4];
        out_cnt = (float)t[5];
        const double nc = t[6], ne = 32.0 * t[6];
        if (nc > 1.0) {
            const double var_c = t[1] / nc - (t[0] / nc) * (t[0] / nc);     // variance of the chunks' sums (32 neighbours each)
            const double var_e = t[3] / ne - (t[2] / ne) * (t[2] / ne);     // variance of the elements
            if (var_e > 0.0) deff = (float)fmin(fmax(var_c / (32.0 * var_e), 1.0), 64.0);
        }
    }
    DPL_PROF_T(qp1);
    DPL_PROF_ADD(0, qp0, qp1);
    // ---- per-bin (count, sum, sum of squares) -> suffix sums (thread t owns the 8 bins below 2047 - 8 t; everything in bins >= j)
    {
        constexpr int kPerT = kLogNB / kThreads;
        const int hi = kLogNB - 1 - (int)tid * kPerT;
        uint32_t c[kPerT], ln = 0u;
        float sm[kPerT], sq[kPerT], ls = 0.0f, lq = 0.0f;
        int my_top = 0;   // the highest sampled bin
#pragma unroll
        for (int q = 0; q < kPerT; ++q) {
            const int b = hi - q;
            const unsigned long long v = b == 0 ? 0ull : packed[b];
            c[q] = (uint32_t)(v >> kPackShift);
            my_top = max(my_top, c[q] ? b : 0);
            sm[q] = (float)bin_sum(v & kPackMask, c[q], b);
            // values of a bin: mean^2 + (bin width)^2 / 12 each
            const float mean = c[q] ? sm[q] / (float)c[q] : 0.0f, wd = log_edge(b + 1) - log_edge(b);
            sq[q] = (float)c[q] * (mean * mean + wd * wd * (1.0f / 12.0f));
            ln += c[q];
            ls += sm[q];
            lq += sq[q];
        }
        uint32_t in = ln;
        float is = ls, iq = lq;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t tn = __shfl_up(in, o, kWave);
            const float ts = __shfl_up(is, o, kWave), tq = __shfl_up(iq, o, kWave);
            if (lane >= (uint32_t)o) {
                in += tn;
                is += ts;
                iq += tq;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_top = max(my_top, __shfl_xor(my_top, o, kWave));
        if (lane == kWave - 1) {
            red_n[w] = in;
            red_s[w] = is;
            red_q[w] = iq;
        }
        if (lane == 0) red_top[w] = my_top;
        __syncthreads();   // every thread has read its packed words: the area may be overwritten
        uint32_t rn = in - ln;
        float rs = is - ls, rq = iq - lq;
        for (int q = 0; q < w; ++q) {
            rn += red_n[q];
            rs += red_s[q];
            rq += red_q[q];
        }
#pragma unroll
        for (int q = 0; q < kPerT; ++q) {
            rn += c[q];
            rs += sm[q];
            rq += sq[q];
            n_ge[hi - q] = rn;
            s_ge[hi - q] = rs;
            q_ge[hi - q] = rq;
        }
    }
    if (tid < (uint32_t)kLogWords) sh.bm[tid] = 0u;
    __syncthreads();
    DPL_PROF_T(qp2);
    DPL_PROF_ADD(1, qp1, qp2);
    // ---- the iteration on the sample, with its uncertainty (one thread: ~20 steps of a few dozen operations)
#ifdef DPL_PROBE_NOWALK
    if (tid == 0) sh.bm[20] = 0xFFu;
#else
    if (tid == 0) {
        const float ud = (dynamic_sym && fabsf(smn) < 1e-6f) ? 4.0f : 1.0f;
        const float c = (float)(1.0 / 65536.0 / 3.0) / ud;
        const float fpc = 1.0f - 1.0f / (float)rate;           // finite population: the sample is a fixed share of the pair
        auto mark = [&](int a, int b) {   // bins a .. b
            a = max(a, 1);
            b = min(b, kLogNB - 2);
            for (int w0 = a >> 5; w0 <= b >> 5; ++w0) {
                const int lo_b = max(a, w0 << 5) & 31, hi_b = min(b, (w0 << 5) + 31) & 31;
                sh.bm[w0] |= (0xFFFFFFFFu >> (31 - hi_b)) & (0xFFFFFFFFu << lo_b);
            }
        };
        const float nz = (float)n_ge[1] + out_cnt;
        if (n_ge[1] > 0u && m_all > 0u) {
            // highest sampled bin + an octave: the pair's maximum lies above the sample's
            const int top = min(max(max(red_top[0], red_top[1]), max(red_top[2], red_top[3])) + 64, kLogNB - 2);
            float s = (s_ge[1] + out_sum) / nz;
            float V = fmaxf(q_ge[1] / nz - s * s, 0.0f) / nz * fpc * deff;
            for (int k = 0; k <= max_iters; ++k) {
                const float sd = sqrtf(V), lo = fmaxf(s - z * sd, 1e-30f), hi = s + z * sd;
                const int jl = max(log_bin(lo), 1), jh = log_bin(hi);
                if (n_ge[jl] < kProbeThin) {   // the sparse tail, wholesale
                    mark(jl, top);
                    break;
                }
                mark(jl, jh);
                const int j = min(max(log_bin(s), 1), kLogNB - 2);
                const float e0 = log_edge(j), e1 = log_edge(j + 1), fr = fminf(fmaxf((e1 - s) / (e1 - e0), 0.0f), 1.0f);
                const float cj = (float)(n_ge[j] - n_ge[j + 1]);
                const float ngt = (float)n_ge[j + 1] + fr * cj;
                const float sgt = s_ge[j + 1] + fr * (s_ge[j] - s_ge[j + 1]);
                const float qgt = q_ge[j + 1] + fr * (q_ge[j] - q_ge[j + 1]);
                if (!(ngt > 0.0f)) {
                    mark(jl, top);
                    break;
                }
                const float s1 = sgt / (c * ((float)m_all - ngt) + ngt);
                const float mean_t = sgt / ngt, var_t = fmaxf(qgt / ngt - mean_t * mean_t, 0.0f);
                const float fp = fminf(fmaxf(cj / (e1 - e0) * (s1 - s) / ngt, 0.0f), 1.0f);   // F'(s) = density (F - s) / N_gt
                V = var_t / ngt * fpc * deff + fp * fp * V;
                if (fabsf(s1 - s) < 1e-6f || !(s1 == s1)) break;
                s = s1;
            }
        }
    }
#endif
    __syncthreads();
    DPL_PROF_T(qp3);
    DPL_PROF_ADD(2, qp2, qp3);
    // ---- the row: at most kMaxFlag - 1 bins (lowest first) + per word the number of gathered bins below it (wave 0)
    if (tid < (uint32_t)kLogWords) {
        uint32_t x = sh.bm[tid];
        const uint32_t incl = wave_incl_scan_dpp((uint32_t)__popc(x));
        uint32_t below = incl - (uint32_t)__popc(x);
        if (incl > (uint32_t)(kMaxFlag - 1)) {
            const uint32_t keep = below < (uint32_t)(kMaxFlag - 1) ? (uint32_t)(kMaxFlag - 1) - below : 0u;
            while ((uint32_t)__popc(x) > keep) x &= ~(1u << (31 - __clz(x)));
            below = min(below, (uint32_t)(kMaxFlag - 1));
        }
        row[tid] = x;
        row[kLogWords + tid] = below;
    }
}

#endif   // DPL_WITH_ONEREAD
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
uint32_t dpl_octav_sort_chunk(void) { return kChunk; }
uint32_t dpl_octav_dir_row(void) { return (uint32_t)kDirRow; }
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

// The round-3 form (job.tail == 0: prediction of the bins all iterates visit, k_octav_oneread / k_octav_probe / k_octav_walk /
// k_octav_sort / k_octav_walk_sorted) is compiled only with -DDPL_WITH_ONEREAD (DPL_WITH_ONEREAD=1 python -m
// dipoorlet_amd.csrc.build): the exact-tail form superseded it in round 4; it is kept for A/B runs.
int dpl_octav_has_oneread(void) {
#ifdef DPL_WITH_ONEREAD
    return 1;
#else
    return 0;
#endif
}
#ifndef DPL_WITH_ONEREAD
static int no_round3(const char* who) {
    snprintf(g_err, sizeof(g_err), "%s: job.tail == 0, but this library was built without the round-3 one-read form (-DDPL_WITH_ONEREAD)", who);
    return -5;
}
#endif

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
    // (the round-3 form's own: prediction rows per pair, the choice per tensor, the selection statistics, the sorted runs)
    if (!j->tail && (!j->d_lh || !j->d_slice_chunk0 || !j->d_dir || !j->d_pred_pair || !j->d_use_probe || !j->d_tstat)) {
        snprintf(g_err, sizeof(g_err), "%s: null buffer in the job (round-3 form)", who);
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

int dpl_octav_oneread_prepare(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_prepare");
    const int64_t vis_words = j->n_tensors * kLogWords;
    uint32_t* d_vis_w = j->d_vis + (int64_t)j->write_epoch * vis_words;
    const uint32_t* d_vis_o = j->d_vis + (int64_t)(1 - j->write_epoch) * vis_words;
    const int64_t init_n = (j->n_pairs + 1 > vis_words ? j->n_pairs + 1 : vis_words);
    if (j->predict < 0 || j->predict > 2) return fail_msg("dpl_octav_oneread_prepare: predict must be 0, 1 or 2");
    if (j->tail) {   // exact-tail form: state + the tensors' threshold snapshot
        const int64_t n = j->n_pairs + 1 > j->n_tensors ? j->n_pairs + 1 : j->n_tensors;
        hipLaunchKernelGGL(k_octav_tail_init, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, j->d_states, j->n_pairs, d_vis_w, d_vis_o,
                           j->d_pred, j->n_tensors, j->reset_epoch);
        DPL_LAUNCH_CHECK("k_octav_tail_init");
        return 0;
    }
#ifdef DPL_WITH_ONEREAD
    hipLaunchKernelGGL(k_octav_oneread_init, dim3(grid_for(init_n, 256)), dim3(256), 0, (hipStream_t)s, j->d_states, j->n_pairs, d_vis_w,
                       d_vis_o, j->d_pred, vis_words, j->reset_epoch, j->d_tstat, j->d_use_probe, j->predict);
    DPL_LAUNCH_CHECK("k_octav_oneread_init");
    return 0;
#else
    (void)init_n;
    return no_round3("dpl_octav_oneread_prepare");
#endif
}

int dpl_octav_oneread_probe(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_probe");
    if (j->tail) return 0;   // the exact-tail form needs no prediction row
#ifndef DPL_WITH_ONEREAD
    return no_round3("dpl_octav_oneread_probe");
#else
    hipLaunchKernelGGL(k_octav_probe, dim3((unsigned)j->n_pairs), dim3(kThreads), 0, (hipStream_t)s, j->d_pair_spans, j->d_seg_ptrs,
                       j->d_pred, j->d_use_probe, j->d_pred_pair, (uint32_t)j->n_tensors, j->dynamic_sym, j->max_iters,
                       j->probe_z > 0.0f ? j->probe_z : kProbeZ, j->d_pair_order, j->probe_z > 0.0f ? nullptr : j->d_tstat);
    DPL_LAUNCH_CHECK("k_octav_probe");
    return 0;
#endif
}

int dpl_octav_oneread_stream(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_stream");
    if (j->tail) {
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
#ifndef DPL_WITH_ONEREAD
    return no_round3("dpl_octav_oneread_stream");
#else
    hipLaunchKernelGGL(k_octav_oneread, dim3((unsigned)j->n_slices), dim3(kThreads), (size_t)(kLdsA + kLdsB), (hipStream_t)s,
                       j->d_slices, j->d_seg_ptrs, j->d_states, reinterpret_cast<unsigned long long*>(j->d_lh), PredRows{j->d_pred, j->d_pred_pair, j->d_use_probe},
                       (uint32_t)j->n_tensors, j->d_pair_base, j->d_pair_slice0, j->d_list0, j->d_states + j->n_pairs,
                       FusedArgs{j->d_vis + (int64_t)j->write_epoch * j->n_tensors * kLogWords, j->d_rescue_bm, j->d_missed, j->d_tstat,
                                 reinterpret_cast<unsigned long long*>(j->d_resc), j->dynamic_sym, j->max_iters, g_exact_fail_every,
                                 j->fuse});
    DPL_LAUNCH_CHECK("k_octav_oneread");
    return 0;
#endif
}

// Everything behind the streaming kernel, in stream order, nothing decided on the host:
//   the walk (k_octav_walk, or k_octav_sort + k_octav_walk_sorted + k_octav_walk for the small pairs and, phase 1, for the pairs
//   the sorted walk could not finish)  ->  the rescue of the pairs whose walk left the gathered bins (k_octav_rescue_gather:
//   those pairs re-read alone for their exact bracket's bins; k_octav_walk phase 2)  ->  the compaction route for what even
//   that could not finish.  Every kernel behind the walk returns at once when the control block says there is nothing for it.
int dpl_octav_oneread_finish(const dpl_octav_oneread_job* j, dpl_stream_t s) {
    DPL_JOB_CHECK("dpl_octav_oneread_finish");
    hipStream_t st = (hipStream_t)s;
    uint32_t* d_vis_w = j->d_vis + (int64_t)j->write_epoch * j->n_tensors * kLogWords;
    const unsigned long long* lh = reinterpret_cast<const unsigned long long*>(j->d_lh);
    dpl_octav_state* ctl = j->d_states + j->n_pairs;
#ifdef DPL_WITH_ONEREAD
    auto walk = [&](unsigned grid, const uint32_t* order, int phase) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_octav_walk<kVec>), dim3(grid), dim3(kThreads), 0, st, j->d_states, ctl, order, lh, j->d_pair_slice0, PredRows{j->d_pred, j->d_pred_pair, j->d_use_probe},
                           d_vis_w, (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_slices, j->dynamic_sym, j->max_iters,
                           g_exact_fail_every, phase, j->d_rescue_bm, j->d_missed, j->d_pred, j->d_tstat,
                           reinterpret_cast<unsigned long long*>(j->d_resc));
    };
    // d_pair_order: largest first — the multi-slice pairs are its first n_multi entries, the small pairs its last n_small.
    // fuse: the streaming kernel has walked every single-slice pair itself; what is left here are the multi-slice ones.
    const int64_t n_big = j->fuse ? j->n_multi : j->n_pairs - j->n_small;
    const int64_t n_small = j->fuse ? 0 : j->n_small;
    const int64_t n_walk = j->fuse ? j->n_multi : j->n_pairs;
    if (j->tail) {
        // every pair was walked by its streaming workgroup; what is left is the rescue below
    } else if (!j->sorted) {   // every pair walked from registers by one workgroup
        if (n_walk > 0 && j->fuse)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_octav_walk<2 * kVec>), dim3((unsigned)n_walk), dim3(kThreads), 0, st, j->d_states, ctl,
                               j->d_pair_order, lh, j->d_pair_slice0, PredRows{j->d_pred, j->d_pred_pair, j->d_use_probe}, d_vis_w,
                               (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_slices, j->dynamic_sym, j->max_iters,
                               g_exact_fail_every, 0, j->d_rescue_bm, j->d_missed, j->d_pred, j->d_tstat,
                               reinterpret_cast<unsigned long long*>(j->d_resc));
        else if (n_walk > 0)
            walk((unsigned)n_walk, j->d_pair_order, 0);
        DPL_LAUNCH_CHECK("k_octav_walk");
    } else {
        if (n_big > 0) {
            hipLaunchKernelGGL(k_octav_sort, dim3((unsigned)j->n_slices), dim3(kThreads), 0, st, j->d_slices, j->d_pair_slice0, lh, PredRows{j->d_pred, j->d_pred_pair, j->d_use_probe},
                               (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_slice_chunk0, j->d_dir, j->fuse);
            DPL_LAUNCH_CHECK("k_octav_sort");
            hipLaunchKernelGGL(k_octav_walk_sorted, dim3((unsigned)n_big), dim3(kWave), 0, st, j->d_states, ctl, j->d_pair_order, lh,
                               j->d_pair_slice0, PredRows{j->d_pred, j->d_pred_pair, j->d_use_probe}, d_vis_w, (uint32_t)j->n_tensors, j->d_pair_base, j->d_list0, j->d_slices,
                               j->d_slice_chunk0, j->d_dir, j->dynamic_sym, j->max_iters, g_exact_fail_every, j->d_pred, j->d_tstat);
            DPL_LAUNCH_CHECK("k_octav_walk_sorted");
        }
        if (n_small > 0) {   // whole window gathered, at most 20 480 values: walked from registers
            walk((unsigned)n_small, j->d_pair_order + n_big, 0);
            DPL_LAUNCH_CHECK("k_octav_walk");
        }
        if (n_big > 0) {        // the pairs the sorted walk marked: their bracket, their place on the rescue list
            walk((unsigned)n_big, j->d_pair_order, 1);
            DPL_LAUNCH_CHECK("k_octav_walk(only_missed)");
        }
    }
#else
    if (!j->tail) return no_round3("dpl_octav_oneread_finish");
    (void)d_vis_w;      // (exact-tail form: every pair was walked by its streaming workgroup; what is left is the rescue below)
#endif
    if (j->max_iters <= 0) return 0;
    if (int e = dpl_octav_rescue_gather_launch(j->d_missed, j->d_states, j->n_pairs, j->d_pair_spans, j->d_seg_ptrs, j->d_rescue_bm,
                                               j->d_pair_base, j->d_list1, st))
        return e;
    hipLaunchKernelGGL(k_octav_walk_rescue, dim3(kRescueGrid), dim3(kThreads), 0, st, j->d_states, ctl, lh, j->d_pair_slice0,
                       (uint32_t)j->n_tensors, j->d_pair_base, j->d_slices, j->dynamic_sym, j->max_iters, g_rescue_fail_every,
                       j->d_rescue_bm, j->d_missed, j->d_list1, reinterpret_cast<unsigned long long*>(j->d_resc));
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
    j->tail = 1;
    j->fuse = 1;
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
    if (int e = dpl_octav_oneread_probe(j, s)) return e;
    if (int e = dpl_octav_oneread_stream(j, s)) return e;
    if (int e = dpl_octav_oneread_finish(j, s)) return e;
    return j && !j->compaction_inline ? dpl_octav_oneread_compaction(j, s) : 0;
}

}  // extern "C"
