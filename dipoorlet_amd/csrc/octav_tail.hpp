// OCTAV ('-A mse', forward_net.py:323-330) in one read: EXACT TAIL, BOUNDED BULK (round 4).  Included by octav_tail_host.hip
// (shares its LDS layout, scans, the walk's counting idiom and the rescue of pairs a walk cannot finish).
//
// The reference's loop s' = sum_{|x|>s} |x| / (c #{|x|<=s} + #{|x|>s}) climbs from s_0 = mean of the non-zero |x| to the
// LEAST fixed point of a step function F, and that fixed point is carried by the top few dozen ... few thousand values of
// the pair: below it F is non-decreasing (dropping a value v raises F iff v < (1 - c) F), so whatever the early iterates
// are — as long as none of them is ABOVE the reference's — the walk ends where the reference ends.  Hence:
//   list    only |x| >= theta = the lower edge of histogram bin J (about 1 % of a pair; round 3 listed the bins of all ~10
//           iterates: 3 - 8 %): a compare on the bin key, no returning atomic, no prediction bitmap;
//   bulk    while the iterate t lies in a bin b < J:  t <- a LOWER BOUND of F(t) from the exact suffix totals alone:
//               F(t) = (S_ge[b+1] + sum of the m values of bin b above t) / (c (n - N_ge[b+1] - m) + N_ge[b+1] + m),
//           every such value lies in (t, edge[b+1]) and 0 <= m <= count[b]; with the values put at t the quotient is monotone
//           in m, so F(t) >= min(q(0), q(count[b])).  By induction a bulk iterate never passes the reference's iterate of the
//           same index;
//   exact   from the first iterate in a bin >= J on: the reference's own step on exact totals (suffix totals of the bins above
//           + the listed values of the iterate's bin, as walk_pair counts them);
//   accept  iff every bulk step moved up by >= 2e-6 (the reference cannot have stopped there), no exact step moved down, the
//           stop rule fired on an exact evaluation, >= 2 exact evaluations, <= max_iters evaluations in all (the reference is
//           never behind: it converged too).  tests/octav_tail_model.py states the same rule in numpy and
//           tests/test_octav_tail_model.py holds it to the oracle over > 10^4 random distributions and thresholds.
//   else    the pair is RESCUED exactly as in round 3: its exact bracket from the histogram, a re-read of that pair alone
//           (k_octav_rescue_gather), the verified walk (k_octav_walk_rescue), the compaction route behind that.
// Where theta comes from: per tensor the LOWEST bin any of its pairs asked for in the last two epochs of batches (a pair asks
// for the bin above which 1/256 of its elements lie: the fixed point of a thin-tailed — uniform — pair has ~0.3 % of the pair
// above it, of a normal one 0.01 %), and it is RAISED ON THE FLY when a wave lists more than its budget (a brighter image, or
// no history at all: the first batch of a run starts at bin 1): that wave alone takes the quantile of the workgroup's
// histogram so far and publishes the new bin in LDS.  theta only ever rises inside a pair, so the list holds every value
// >= the final theta; a theta that ends up too high costs a rescue, never a wrong result.
#pragma once

#ifndef DPL_TAIL_TAU_SHIFT
#define DPL_TAIL_TAU_SHIFT 8      // a pair asks for the bin above which n >> 8 of its n elements lie (measured: 7 -> 8 -2 %, 9 the same with four times the rescues at +-30 %)
#endif
#ifndef DPL_TAIL_BUDGET_SHIFT
#define DPL_TAIL_BUDGET_SHIFT 6   // a wave may list kTailAllow0 + (elements it has seen >> 6) values before it raises theta
#endif
#ifndef DPL_TAIL_ALLOW0
#define DPL_TAIL_ALLOW0 2048
#endif
#ifndef DPL_TAIL_OCC
#define DPL_TAIL_OCC 4
#endif
#ifndef DPL_TAIL_QUEUE_CAP
#define DPL_TAIL_QUEUE_CAP 512    // entries of a wave's survivor queue (flushed above 256): 8 KiB per workgroup, which the walk's suffix counts reuse
#endif
#ifndef DPL_TAIL_VEC
#define DPL_TAIL_VEC 12           // 16-byte vectors per thread the walk keeps the list in (1024 values each); longer lists are streamed from L2
#endif
constexpr int kTailQueueCap = DPL_TAIL_QUEUE_CAP;
constexpr int kTailLdsB = kWaves * kTailQueueCap * 4;
static_assert(kTailQueueCap >= 512 && kTailLdsB >= kLogNB * 4, "a vector may add 256 survivors past the flush mark; the walk keeps its suffix counts in the queues' space");
constexpr int kTailVec = DPL_TAIL_VEC;
constexpr int kTailTauShift = DPL_TAIL_TAU_SHIFT;
constexpr int kTailBudgetShift = DPL_TAIL_BUDGET_SHIFT;
constexpr uint32_t kTailAllow0 = DPL_TAIL_ALLOW0;

#ifdef DPL_RES_PROF
// (stamps taken inside a branch are kept in registers and added at the end: an add is a global read-modify-write, and one in the
// middle of the walk would be measured by the next stamp)
#define DPL_PROF_KEEP(slot, a, b) prof_keep[slot] = (b) - (a)
#define DPL_PROF_FLUSH() do { if (threadIdx.x == 0) { g_res_prof[(blockIdx.x & 4095u) * 8 + 1] += prof_keep[1]; g_res_prof[(blockIdx.x & 4095u) * 8 + 2] += prof_keep[2]; } } while (0)
#else
#define DPL_PROF_KEEP(slot, a, b) do {} while (0)
#define DPL_PROF_FLUSH() do {} while (0)
#endif

struct TailArgs {
    uint32_t* vis_w;             // [T, kLogWords]: word 0 of a tensor's row = kLogNB - (lowest bin its pairs asked for this epoch); 0: none
    const uint32_t* pred;        // [T, kPredRow]: word 0 = the snapshot (both epochs); 0: no history
    uint32_t* rescue_bm;
    uint32_t* missed;
    unsigned long long* resc;
    int dynamic_sym, max_iters, fail_every;
};

// One pair, streamed: min / max, the exact log-scale histogram (ONE non-returning 64-bit LDS add per element) and the values
// at or above the current threshold bin -> the wave's dense LDS queue -> the pair's list.  Leaves the per-wave ranges in
// sh.red_*, the number of listed values in sh.cursor, the final threshold bin in sh.tail_j.
// (Inlined: as a function of its own — round 3's stream_slice — its prologue saves two dozen callee-saved registers per lane to
// scratch and restores them at the end: 97 MB written and 97 MB read per ResNet-50 batch, found in the WRITE_SIZE counter.)
#ifdef DPL_TAIL_NOINLINE
__device__ __attribute__((noinline)) void stream_tail(
#else
__device__ __forceinline__ void stream_tail(
#endif
const float* __restrict__ pg, uint32_t cnt, uint32_t* __restrict__ dst,
                                                      Shared& sh, dpl_octav_state* __restrict__ ctl, const bool adaptive) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const lptr_u64 l_packed = (lptr_u64)(lds_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    float mn = INFINITY, mx = -INFINITY;
    const lptr_u32 wq = (lptr_u32)(lds_raw + kLdsA) + (uint32_t)w * kTailQueueCap;
    uint32_t tail = 0u;          // entries in the wave's queue
    uint32_t mine = 0u;          // values this wave has listed
    uint32_t seen = 0u;          // elements this wave has consumed
    // what the wave may list before it looks at the histogram: 1 / 128 of its share of the pair, at least 256 values (a first
    // look needs something to look at), at most kTailAllow0
    uint32_t slack = min(max(cnt >> 9, 256u), kTailAllow0);
    uint32_t backoff = 512u;     // a raise that could not move the threshold (an atom at the top: saturating activations) doubles it
    uint32_t raises = 0u;
    int jm1 = (int)__builtin_amdgcn_readfirstlane((int)sh.tail_j) - 1;   // listed: bin key t = bin - 1 >= jm1
    // what the list region holds (list_cap_of): a flush that would pass it is dropped, the cursor still counts it — a list
    // longer than its region says "not all listed" and the pair's walk is refused (walk_tail, k_octav_tail_merge)
    const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.list_cap);
    auto flush = [&]() {
        typedef __attribute__((address_space(1))) uint32_t* gptr_u32;
        gptr_u32 gdst = (gptr_u32)dst;
        uint32_t base = 0u;
        if (lane == 0) base = __hip_atomic_fetch_add((lptr_u32)&sh.cursor, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base + tail <= cap)
            for (uint32_t i = lane; i < tail; i += kWave) gdst[base + i] = wq[i] & 0x7FFFFFFFu;
        tail = 0u;
    };
    // The threshold of a wave that lists too much: the bin above which 1 / 2^kTailTauShift of what the workgroup has seen so
    // far lies, from a snapshot of the LDS histogram (the other waves keep adding to it: nothing here has to be exact).  Lane l
    // owns the 32 bins below kLogNB - 32 l (lane 0: the highest).
    auto raise = [&]() {
        const uint32_t b0 = (uint32_t)kLogNB - 32u * (lane + 1u);
        uint32_t tot = 0u;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) tot += (uint32_t)(l_packed[b0 + (uint32_t)k] >> kPackShift);
        const uint32_t incl = scan_u32_dpp(tot);   // everything in bins >= the lane's lowest
        const uint32_t target = max((4u * seen) >> kTailTauShift, 16u);
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(incl >= target);
        int jn = 1;
        if (mm != 0ull) {
            const int L = (int)__builtin_ctzll(mm);   // the first lane (highest bins) whose cumulative count reaches the target
            uint32_t run = incl - tot, found = 0u;
            int jl = (int)b0;
#pragma unroll 8
            for (int k = 31; k >= 0; --k) {
                run += (uint32_t)(l_packed[b0 + (uint32_t)k] >> kPackShift);
                const bool hit = !found && run >= target;
                jl = hit ? (int)b0 + k : jl;
                found |= hit ? 1u : 0u;
            }
            jn = __builtin_amdgcn_readlane(jl, L);
        }
        jn = min(max(jn, 1), kLogNB - 2);
        if (jn - 1 > jm1) {
            if (lane == 0) atomicMax(&sh.tail_j, (uint32_t)jn);
            jm1 = jn - 1;
        } else {
            backoff = min(backoff * 2u, 1u << 20);
        }
        ++raises;
    };
    uint32_t rare = 0u;
    constexpr uint32_t kWin = (uint32_t)(kLogNB - 1);
    const lptr_u64 dummy = l_packed + kLogNB + lane;
    // per element: the bin key (14 bits of exponent and top mantissa, relative to the window), ONE LDS add {count += 1, mantissa
    // sum += 23 explicit bits} on the bin's word (zeros and values outside the window: the lane's dummy word), and the threshold
    // test on the key itself (signed: below the window is negative; above it — values >= 2^14, inf, NaN — passes and is harmless:
    // such a pair leaves for the compaction route anyway)
    auto add1 = [&](uint32_t bits) -> bool {
        const uint32_t t = ((bits >> kLogShift) & 0x3FFFu) - (kLogKey0 + 1u);
        const bool in = t < kWin;
        const lptr_u64 slot = in ? l_packed + t + 1u : dummy;
        rare |= in ? 0u : bits;
#if !defined(DPL_TAIL_ABL_NOHIST)   // (ablation builds, timing only: scripts/tail_ablate.sh)
        (void)__hip_atomic_fetch_add(slot, (1ull << kPackShift) | (unsigned long long)(bits & 0x7FFFFFu), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        asm volatile("" ::"v"(slot));
#endif
#if defined(DPL_TAIL_ABL_NOLIST)
        return false;
#else
        return (int32_t)t >= jm1;
#endif
    };
    auto put = [&](uint32_t bits, bool f) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        if (m != 0ull) {   // (uniform; about every other element slot of a wave at 1 % listed)
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail));
            if (f) wq[pos] = bits;
            const uint32_t k = (uint32_t)__builtin_popcountll(m);
            tail += k;
            mine += k;
        }
    };
    uint32_t rare_n = 0u;
    for_each_tile<kThreads>(pg, cnt, [&](const f4 (&t)[4], uint32_t base, bool full) {
        if (tail > (uint32_t)(kTailQueueCap - 256)) {   // the regular flush: BEFORE the tile is consumed, AFTER all of it has arrived (stream_slice)
            asm volatile("" ::"v"(t[3].w));
            flush();
        }
        // the workgroup's threshold (another wave may have raised it), and this wave's own budget
        jm1 = max(jm1, (int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((lptr_u32)&sh.tail_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - 1);
        if (adaptive && mine > slack + (seen >> kTailBudgetShift)) {   // (a small pair lists its whole window: no budget)
            raise();
            const uint32_t base_allow = seen >> kTailBudgetShift;
            slack = max(slack, mine + backoff > base_allow ? mine + backoff - base_allow : 0u);
        }
        seen += full ? 1024u : min(1024u, cnt - base);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (full) {
                mn = fminf(mn, fminf(fminf(t[u].x, t[u].y), fminf(t[u].z, t[u].w)));
                mx = fmaxf(mx, fmaxf(fmaxf(t[u].x, t[u].y), fmaxf(t[u].z, t[u].w)));
            } else {   // padding is +0.0: in no histogram bin, never listed; only min / max must skip it
                const uint32_t e = base + (uint32_t)u * 256u + lane * 4u;
                if (e + 0 < cnt) mn = fminf(mn, t[u].x), mx = fmaxf(mx, t[u].x);
                if (e + 1 < cnt) mn = fminf(mn, t[u].y), mx = fmaxf(mx, t[u].y);
                if (e + 2 < cnt) mn = fminf(mn, t[u].z), mx = fmaxf(mx, t[u].z);
                if (e + 3 < cnt) mn = fminf(mn, t[u].w), mx = fmaxf(mx, t[u].w);
            }
            const uint32_t b0 = __float_as_uint(t[u].x), b1 = __float_as_uint(t[u].y), b2 = __float_as_uint(t[u].z), b3 = __float_as_uint(t[u].w);
            const bool f0 = add1(b0), f1 = add1(b1), f2 = add1(b2), f3 = add1(b3);
            put(b0, f0);
            put(b1, f1);
            put(b2, f2);
            put(b3, f3);
            if (tail > (uint32_t)(kTailQueueCap - 256)) flush();   // (a pair that lists most of what it reads: a small pair, a cold start)
#ifdef DPL_TAIL_FENCE
            DPL_SCHED_FENCE();
#endif
        }
        if (__any((rare & 0x7FFFFFFFu) != 0u)) {   // non-zero values outside the window (and NaNs): summed directly, as stream_slice does
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
    if (lane == 0) {
        if (rare_n != 0u) atomicAdd(&ctl->reserved, rare_n);   // (statistics: tiles holding values outside the window)
        if (raises != 0u) atomicAdd(&ctl->iters, raises);      // (statistics: thresholds raised on the fly)
    }
    const float wmn = wave_min(mn), wmx = wave_max(mx);
    if (lane == 0) {
        sh.red_mn[w] = wmn;
        sh.red_mx[w] = wmx;
    }
}

// The walk of one pair by the workgroup that has just streamed it.  What it costs is the time a streaming slot stands still and
// the VALU cycles it takes from the streaming waves of the same SIMDs (every instruction of a walk that all four waves execute is
// paid four times: measured, the round-3 shape spent 100 us of every SIMD per batch on walks), so:
//   * ONE wave walks; the others only join for what is parallel — the totals above every group of 8 bins, the compaction of a
//     long list — and otherwise wait at a barrier, which costs no issue slot;
//   * the histogram stays PACKED in LDS: per group of 8 bins (one thread) the totals above the group (3 KiB); the suffix totals at a
//     bin are looked up lazily — eight lanes read the group's eight words, a DPP sum — about fifteen times per walk, instead of
//     converting all 2048 bins to fp64 suffix totals up front;
//   * bounded steps go on past the list's first bin until the values still above the iterate fit one wave's registers (kSurvCap);
//     they are then compacted through LDS once (a list that fits is loaded that way to begin with): an exact step is 20 compares
//     per lane and a DPP sum, no exchange, no barrier;
//   * lists beyond that (a cold start, a pair much brighter than its tensor's history) take the round-3 shape: rows spread over
//     the workgroup, partial sums exchanged through LDS.
#ifndef DPL_TAIL_SURV_VEC
#define DPL_TAIL_SURV_VEC 5
#endif
constexpr int kSurvVec = DPL_TAIL_SURV_VEC;          // 16-byte vectors per lane a wave holds the surviving values in
constexpr uint32_t kFitCap = kSurvVec * 4 * kWave;   // values one wave's registers hold (a list this short is loaded there whole)
// ... and what the compaction's staging area holds (group totals, 3 KiB, + the staging area share the queues' space)
constexpr uint32_t kSurvCap = kFitCap * 4 <= (uint32_t)(kTailLdsB - 3072) ? kFitCap : (uint32_t)(kTailLdsB - 3072) / 4;

template <int kVecT>
__device__ __forceinline__ void walk_tail(const uint32_t pair, const uint32_t tensor, unsigned char* lds_raw, Shared& sh,
                                          dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl,
                                          const uint64_t* __restrict__ pair_base, const float* __restrict__ list0,
                                          const TailArgs& fa, const uint32_t cnt, const unsigned long long n_merged = 0ull) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    dpl_octav_state* me = st + pair;
    const unsigned long long n_pair = n_merged ? n_merged : (unsigned long long)cnt;   // (n_merged: a pair of several slices, k_octav_tail_merge)
    const bool small = !n_merged && cnt <= kSmallCap;
#ifdef DPL_RES_PROF
    unsigned long long prof_keep[3] = {0ull, 0ull, 0ull};
#endif
    const unsigned long long* packed = reinterpret_cast<const unsigned long long*>(lds_raw);   // the histogram: intact to the end
    uint32_t* tn = reinterpret_cast<uint32_t*>(lds_raw + kLdsA);                                // [256] counts above a group
    double* ts = reinterpret_cast<double*>(lds_raw + kLdsA + 1024);                             // [256] sums above a group
    uint32_t* surv = reinterpret_cast<uint32_t*>(lds_raw + kLdsA + 3072);                       // [kSurvCap] survivors' staging
    // the list: this workgroup's own global stores (one CU, one L1), requested before anything else
    const float* lp = list0 + pair_base[pair];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t L_listed = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.cursor);
    // more listed than the region holds: part of the list was dropped (stream_tail's flush) — nothing of it is read, the walk refused
    const bool over = L_listed > (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.list_cap);
    const uint32_t L = over ? 0u : L_listed;
    const uint32_t n_rows = (L + 1023u) >> 10;
    const bool fits = L <= kFitCap;
    // ONE register array, two layouts: the workgroup's rows of 1024 values (a list that does not fit a wave), or — in its first
    // kSurvVec vectors, wave 0 only — all surviving values (entry (u * 64 + lane) * 4 ...)
    f4 v[kVecT];
    static_assert(kVecT >= kSurvVec, "the survivors live in the rows' registers");
    auto load_rows = [&](auto& dst, auto count, uint32_t row0) {
        constexpr int kN = decltype(count)::value;
        const uint32_t voff = tid << 4;
#pragma unroll
        for (int u = 0; u < kN; ++u) {
            const uint32_t e0 = (row0 + (uint32_t)u) << 10;
            const int nbytes = e0 < L ? (int)(min(L - e0, 1024u) << 2) : 0;   // buffer loads: zero fill past the list's end
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(lp + (e0 < L ? e0 : 0u)), 0, nbytes, 0x00020000);
            dst[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        }
    };
    if (fits) {
        if (w == 0) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)lp, 0, (int)(L << 2), 0x00020000);
#pragma unroll
            for (int u = 0; u < kSurvVec; ++u)
                v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((uint32_t)u * kWave + lane) << 4, 0, 0));
        }
    } else {
        load_rows(v, std::integral_constant<int, kVecT>{}, 0u);
    }
    DPL_PROF_T(wt0);
    // ---- totals above every group of 8 bins (thread t: bins 2040 - 8 t .. 2047 - 8 t, one exponent: integer sums, ONE conversion)
    constexpr int kPerT = kLogNB / kThreads;
    static_assert(kPerT == 8, "a thread owns eight bins: an eighth of an octave");
    const int hi = kLogNB - 1 - (int)tid * kPerT;
    {
        uint32_t c[kPerT], cn = 0u;
        unsigned long long cm = 0ull;
#pragma unroll
        for (int q = 0; q < kPerT; ++q) {
            const int bq = hi - q;
            const unsigned long long x = bq == 0 ? 0ull : packed[bq];   // bin 0 holds no element
            c[q] = (uint32_t)(x >> kPackShift);
            cn += c[q];
            cm += (x & kPackMask) + ((unsigned long long)c[q] << 23);   // full 24-bit mantissas
        }
        const double sd = (double)cm * log_bin_scale(hi);
        const uint32_t in = scan_u32_dpp(cn);      // threads ascending = bins descending: everything in the groups at and above mine
        const double is = scan_f64_dpp(sd);
        if (lane == kWave - 1) {
            sh.red_a[w] = in;
            sh.red_d[w] = is;
        }
        if (tid == 0) sh.jwant = 1u;
        __syncthreads();
        uint32_t above_n = in - cn;
        double above_s = is - sd;
        for (int q = 0; q < w; ++q) {
            above_n += sh.red_a[q];
            above_s += sh.red_d[q];
        }
        tn[tid] = above_n;
        ts[tid] = above_s;
        // what this pair asks the tensor's next batches to list: the bin above which n >> kTailTauShift elements lie
        const uint32_t want = max((uint32_t)(n_pair >> kTailTauShift), 1u);
        uint32_t run = above_n;
#pragma unroll
        for (int q = 0; q < kPerT; ++q) {
            const uint32_t prev = run;
            run += c[q];
            if (hi - q >= 1 && run >= want && prev < want) sh.jwant = (uint32_t)(hi - q);   // (N_ge is monotone: one bin at most)
        }
        __syncthreads();
    }
    // The suffix totals at bin b, lazily (wave-uniform b): N_ge[b + 1], S_ge[b + 1] and the bin's own count.  Lanes 0 .. 7 read
    // the eight words of b's group, masked to the bins above b; a DPP sum over the eight lanes.
    uint32_t lk_n = 0u, lk_c = 0u;
    double lk_s = 0.0;
    auto look = [&](int b) {
        const int t = (kLogNB - 1 - b) >> 3, g0 = kLogNB - kPerT - 8 * t;
        const int bq = g0 + (int)(lane & 7u);
        const unsigned long long x = bq == 0 ? 0ull : packed[bq];
        const uint32_t c1 = (uint32_t)(x >> kPackShift);
        const unsigned long long m1 = (x & kPackMask) + ((unsigned long long)c1 << 23);
        uint32_t ci = bq > b ? c1 : 0u;
        unsigned long long mi = bq > b ? m1 : 0ull;
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane((int)c1, b - g0);
        // sum over lanes 0 .. 7 of the row (every group of eight lanes holds the same values): quad swaps + half-row mirror
        uint32_t mlo = (uint32_t)mi & 0xFFFFFFu, mhi = (uint32_t)(mi >> 24);   // (two 24-bit halves: eight of them stay below 2^32)
#define DPL_LK_STEP(ctrl)                                                                  \
        ci += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ci, ctrl, 0xF, 0xF, true);     \
        mlo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, ctrl, 0xF, 0xF, true);   \
        mhi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, ctrl, 0xF, 0xF, true);
        DPL_LK_STEP(0xB1)    // quad_perm [1,0,3,2]
        DPL_LK_STEP(0x4E)    // quad_perm [2,3,0,1]
        DPL_LK_STEP(0x141)   // row_half_mirror
#undef DPL_LK_STEP
        ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
        const unsigned long long mt = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mlo) +
                                      ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mhi) << 24);
        lk_n = tn[t] + ci;
        lk_s = ts[t] + (double)mt * log_bin_scale(b);
        lk_c = cb;
    };
    float s = 0.0f, s0 = 0.0f, ud = 1.0f;
    uint32_t route = 2u, bad = 0u, evals = 0u, exact = 0u, stopped = 0u, path = 0u;
    int jb = 0;
    unsigned long long n_above = 0ull;
    double s_above = 0.0;
    double c = 0.0;
    const int max_iters = fa.max_iters;
    // one exact step from the count / mantissa sum of the values of bin jb above s; false: the walk is over
    auto step = [&](unsigned long long tc, unsigned long long tm) {
        const unsigned long long tg = n_above + tc;
        const double tsum = s_above + (double)(tm + (tc << 23)) * log_bin_scale(jb);
        const double denom = c * (double)(long long)(n_pair - tg) + (double)(long long)tg;
        const float s1 = __fdiv_rn((float)tsum, (float)denom);
        ++evals;
        ++exact;
        if (fabsf(__fsub_rn(s1, s)) < 1e-6f) {
            stopped = 1u;   // forward_net.py:328-329: keeps the previous s
            return false;
        }
        if (!(s1 >= s)) {
            bad = 1u;       // moved down (or NaN): not the climb this form relies on
            return false;
        }
        s = s1;
        if ((int)evals >= max_iters) {
            bad = 1u;       // the cap: which iterate the reference stopped on is not known here
            return false;
        }
        const int jn = log_bin(s);
        if (jn > kLogNB - 2) {
            bad = 1u;
            return false;
        }
        if (jn != jb) {
            jb = jn;
            look(jb);
            n_above = (unsigned long long)lk_n;
            s_above = lk_s;
        }
        return true;
    };
    // the exact steps of ONE wave over the survivors in its first kSurvVec vectors
    auto walk_alone = [&]() {
        for (;;) {
            const uint32_t lo1 = __float_as_uint(s) + 1u;
            const uint32_t span = (((uint32_t)(jb + 1) + kLogKey0) << kLogShift) - lo1;
            uint32_t cc = 0u, ds = 0u;   // cc: wave-uniform
#pragma unroll
            for (int u = 0; u < kSurvVec; ++u) {
                const uint32_t b4[4] = {__float_as_uint(v[u].x), __float_as_uint(v[u].y), __float_as_uint(v[u].z), __float_as_uint(v[u].w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t d = b4[e] - lo1;   // values of bin jb above s: d below `span` (anything at or below s wraps around)
                    const bool in = d < span;
                    cc += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
                    ds += in ? d : 0u;
                }
            }
            // (per lane at most 20 values below 2^17: the DPP sum of ds stays below 2^32)
            const unsigned long long tm = (unsigned long long)wave_sum_dpp(ds) + (unsigned long long)cc * (unsigned long long)(lo1 & 0x7FFFFFu);
            if (!step((unsigned long long)cc, tm)) break;
        }
    };
    DPL_PROF_T(wt1);
#ifdef DPL_RES_PROF
    unsigned long long wt2m = wt1;   // (end of wave 0's bounded steps)
#endif
    if (w == 0) {
        // ---- the pair's statistics (stream_tail left them per wave), s_0, the route
        const float gmn = fminf(fminf(sh.red_mn[0], sh.red_mn[1]), fminf(sh.red_mn[2], sh.red_mn[3]));
        const float gmx = fmaxf(fmaxf(sh.red_mx[0], sh.red_mx[1]), fmaxf(sh.red_mx[2], sh.red_mx[3]));
        const bool nanseen = sh.low_nan != 0u;
        // forward_net.py:319 — np.abs(data_min - 0) < 1e-6 (float32 compare) and 'dynamic_sym' in qi_params
        ud = (fa.dynamic_sym && fabsf(gmn) < 1e-6f && !nanseen) ? 4.0f : 1.0f;
        c = 1.0 / 65536.0 / 3.0 / (double)ud;
        look(0);   // N_ge[1], S_ge[1]: the whole window
        // forward_net.py:324 — sum(|x|) / count(|x| > 0): exact window totals + the out-of-window part
        s0 = nanseen ? __uint_as_float(0x7FC00000u)
                     : __fdiv_rn((float)(sh.low_sum + lk_s), (float)(long long)((unsigned long long)sh.low_cnt + lk_n));
        if (s0 != s0 || max_iters <= 0) route = 0u;               // 0: finished (NaN is a fixed point)
        else if (!(fmaxf(fabsf(gmn), fabsf(gmx)) < log_edge(kLogNB))) route = 1u;   // 1: values >= 2^14 / inf: compaction route
        bad = route == 1u ? 1u : 0u;
        s = s0;
        if (route == 2u) {
            const int J = (int)__builtin_amdgcn_readfirstlane((int)sh.tail_j);
            if (fa.fail_every > 0 && pair % (uint32_t)fa.fail_every == 0u) bad = 1u;   // test hook: the rescue path
            if (over) bad = 1u;
            // ---- bounded steps: lower bounds of F from the suffix totals; past the list's first bin while more values lie above
            // the iterate than a wave holds
            jb = log_bin(s);
            while (!bad) {
                if (jb < 1 || jb > kLogNB - 2) {
                    bad = 1u;   // outside the binned window
                    break;
                }
                look(jb);
                const uint32_t nb1 = lk_n, nb = lk_n + lk_c;
                if (jb >= J && (fits || nb <= kSurvCap)) break;   // exact from here on (lk_* hold the totals above bin jb)
                bool moved = false;
                float lb = s;
                if (nb1 != 0u) {
                    const double d0 = c * (double)(long long)(n_pair - nb1) + (double)nb1;
                    const double dm = c * (double)(long long)(n_pair - nb) + (double)nb;
                    const double nm = lk_s + (double)lk_c * (double)s;
                    const float q0 = __fdiv_rn((float)lk_s, (float)d0), qm = __fdiv_rn((float)nm, (float)dm);
                    lb = __fmul_rn(fminf(q0, qm), 0.99999952316284180f);   // (1 - 2^-21: below every rounding above)
                    moved = __fsub_rn(lb, s) >= 2e-6f;
                }
                if (!moved) {
                    if (jb < J) bad = 1u;   // below the list: the fixed point is down here, or the pair is degenerate
                    break;                  // inside it: the bound has stopped moving, exact steps take over (the long way)
                }
                s = lb;
                ++evals;
                if ((int)evals >= max_iters) {
                    bad = 1u;
                    break;
                }
                jb = log_bin(s);
            }
            n_above = (unsigned long long)lk_n;
            s_above = lk_s;
            // 1: this wave finishes alone (the list fits its registers); 2: after a compaction; 3: the workgroup, rows + exchange
            if (!bad) path = fits ? 1u : ((lk_n + lk_c <= kSurvCap && n_rows <= (uint32_t)kVecT) ? 2u : 3u);
#ifdef DPL_RES_PROF
            wt2m = __builtin_readcyclecounter();
#endif
            if (path == 1u) walk_alone();
        }
        if (lane == 0) {
            sh.t_s = s;
            sh.w_s0 = s0;
            sh.w_ud = ud;
            sh.w_jb = jb;
            sh.w_evals = evals;
            sh.w_path = path;
            sh.w_bad = bad;
            sh.w_route = route;
            sh.w_lkn = lk_n;
            sh.w_lkc = lk_c;
            sh.w_lks = lk_s;
        }
    }
    __syncthreads();   // (the other waves have been waiting here: no issue slot spent)
    path = sh.w_path;
    if (path == 2u) {
        // ---- the values still above the iterate (bins >= jb), compacted through LDS by the whole workgroup: count, offsets, place
        const uint32_t need = sh.w_lkn + sh.w_lkc;
        const uint32_t edge = ((uint32_t)sh.w_jb + kLogKey0) << kLogShift;
        uint32_t my = 0u;
#pragma unroll
        for (int u = 0; u < kVecT; ++u) {
            if ((uint32_t)u < n_rows) {   // uniform
                my += (__float_as_uint(v[u].x) >= edge ? 1u : 0u) + (__float_as_uint(v[u].y) >= edge ? 1u : 0u) +
                      (__float_as_uint(v[u].z) >= edge ? 1u : 0u) + (__float_as_uint(v[u].w) >= edge ? 1u : 0u);
            }
        }
        const uint32_t in = scan_u32_dpp(my);
        if (lane == kWave - 1) sh.part_c[0][w] = in;
        __syncthreads();
        uint32_t pos = in - my, total = 0u;
        for (int q = 0; q < kWaves; ++q) {
            pos += q < w ? sh.part_c[0][q] : 0u;
            total += sh.part_c[0][q];
        }
        const bool ok = total == need;   // (else the list does not hold what the histogram counted: cannot happen; rescued)
        if (ok) {
#pragma unroll
            for (int u = 0; u < kVecT; ++u) {
                if ((uint32_t)u < n_rows) {
                    const uint32_t b4[4] = {__float_as_uint(v[u].x), __float_as_uint(v[u].y), __float_as_uint(v[u].z), __float_as_uint(v[u].w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (b4[e] >= edge) surv[pos++] = b4[e];
                    }
                }
            }
            for (uint32_t i = total + tid; i < kSurvCap; i += kThreads) surv[i] = 0u;
        }
        __syncthreads();
        if (w == 0) {
            if (!ok) {
                bad = 1u;
            } else {
#pragma unroll
                for (int u = 0; u < kSurvVec; ++u) v[u] = *reinterpret_cast<const f4*>(surv + (((uint32_t)u * kWave + lane) << 2));
                walk_alone();
            }
            if (lane == 0) {
                sh.t_s = s;
                sh.w_evals = evals;
                sh.w_bad = bad;
            }
        }
        __syncthreads();
    } else if (path == 3u) {
        // ---- rows spread over the workgroup, partial sums through LDS, every wave takes the step (walk_pair's loop)
        s = sh.t_s;
        s0 = sh.w_s0;
        ud = sh.w_ud;
        c = 1.0 / 65536.0 / 3.0 / (double)ud;
        jb = sh.w_jb;
        evals = sh.w_evals;
        n_above = (unsigned long long)sh.w_lkn;
        s_above = sh.w_lks;
        bad = 0u;
        uint32_t par = 0u;
        f4 ov[kOver];
        for (;;) {
            const uint32_t lo1 = __float_as_uint(s) + 1u;
            const uint32_t span = (((uint32_t)(jb + 1) + kLogKey0) << kLogShift) - lo1;
            uint32_t cc = 0u;   // (wave-uniform)
            unsigned long long dsum = 0ull;
            uint32_t ds = 0u;
            auto in1 = [&](float f) {
                const uint32_t d = __float_as_uint(f) - lo1;
                const bool in = d < span;
                cc += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
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
                dsum += (unsigned long long)wave_sum_dpp(ds);
                ds = 0u;
            }
            for (uint32_t r0 = (uint32_t)kVecT; r0 < n_rows; r0 += (uint32_t)kOver) {   // a list beyond the registers
#pragma unroll
                for (int u = 0; u < kOver; ++u) {
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
                sh.part_c[par][w] = cc;
                sh.part_m[par][w] = dsum + (unsigned long long)cc * (unsigned long long)(lo1 & 0x7FFFFFu);
            }
            __syncthreads();
            unsigned long long tc = 0ull, tm = 0ull;
#pragma unroll
            for (int j = 0; j < kWaves; ++j) {
                tc += sh.part_c[par][j];
                tm += sh.part_m[par][j];
            }
            par ^= 1u;
            if (!step(tc, tm)) break;
        }
        if (tid == 0) {
            sh.t_s = s;
            sh.w_evals = evals;
            sh.w_bad = bad;
        }
        // (exact / stopped of this path are every wave's own: the acceptance below is taken by wave 0 from its copies)
        __syncthreads();
    }
    // ---- acceptance (wave 0 holds exact / stopped of paths 1 and 2; every wave its own of path 3)
    s = sh.t_s;
    s0 = sh.w_s0;
    ud = sh.w_ud;
    evals = sh.w_evals;
    route = sh.w_route;
    bad = sh.w_bad;
    if (w == 0 && lane == 0) {
        // at least two exact evaluations behind a bounded start (none needed when every step was exact: a small pair)
        if (route == 2u && !bad && !(stopped && (exact >= 2u || evals == exact))) bad = 1u;
        sh.w_bad = bad;
    }
    __syncthreads();
    bad = sh.w_bad;
    DPL_PROF_T(wt3);
    DPL_PROF_KEEP(1, wt1, wt2m);
    DPL_PROF_KEEP(2, wt2m, wt3);
    DPL_PROF_T(wt4);
    // ---- a pair this form could not finish is RESCUED (as walk_pair does): exact bracket, re-read of the pair alone, verified
    // walk.  The bracket walk wants the suffix totals of every bin: only now are they written out (over the packed histogram)
    bool rescued = false;
    if (bad && !small && route == 2u) {
        double* s_ge = reinterpret_cast<double*>(lds_raw);
        uint32_t* n_ge = reinterpret_cast<uint32_t*>(lds_raw + kLdsA);
        {
            uint32_t c8[kPerT];
            unsigned long long m8[kPerT];
#pragma unroll
            for (int q = 0; q < kPerT; ++q) {
                const unsigned long long x = packed[hi - q];
                c8[q] = (uint32_t)(x >> kPackShift);
                m8[q] = x & kPackMask;
            }
            __syncthreads();   // (n_ge overwrites the group totals and the staging area: every look is behind us)
#pragma unroll
            for (int q = 0; q < kPerT; ++q) {
                const int bq = hi - q;
                if (bq == 0) c8[q] = 0u, m8[q] = 0ull;
                n_ge[bq] = c8[q];
                s_ge[bq] = bin_sum(m8[q], c8[q], bq);
            }
        }
        suffix_in_place(n_ge, s_ge, sh);
        if (tid < (uint32_t)kLogWords) sh.pub[tid] = 0u;
        __syncthreads();
        if (tid == 0) {
            sh.route = bracket_marks(n_ge, s_ge, sh.pub, s0, ud, n_pair).route;
            sh.would_list = 0u;
        }
        __syncthreads();
        // the rescue gathers the values of the bracket's (marked) bins into the pair's region of the rescue list: a bracket that
        // holds more than the region does goes straight to the compaction route
        if (sh.route == 2u) {
            uint32_t held = 0u;
#pragma unroll
            for (int q = 0; q < kPerT; ++q) {
                const int b = hi - q;
                if (b >= 1 && b < kLogNB - 1 && ((sh.pub[b >> 5] >> (b & 31)) & 1u)) held += n_ge[b] - n_ge[b + 1];
            }
            if (held) atomicAdd(&sh.would_list, held);
        }
        __syncthreads();
        if (tid == 0 && sh.route == 2u && sh.would_list > sh.region_cap) sh.route = 1u;
        __syncthreads();
        rescued = sh.route == 2u;
        if (rescued) {
            if (tid < (uint32_t)kLogWords) fa.rescue_bm[(uint64_t)pair * kLogWords + tid] = sh.pub[tid];
            double* rs = reinterpret_cast<double*>(fa.resc + (uint64_t)pair * kRescRow);
            uint32_t* rn = reinterpret_cast<uint32_t*>(fa.resc + (uint64_t)pair * kRescRow + kLogNB);
            for (int b = tid; b < kLogNB; b += kThreads) {
                rs[b] = s_ge[b];
                rn[b] = n_ge[b];
            }
        }
    }
    if (tid == 0) {
        // the pair's state: its statistics (what the rescue / the compaction route / dpl_octav_finalize read), and what became of it
        const float gmn = fminf(fminf(sh.red_mn[0], sh.red_mn[1]), fminf(sh.red_mn[2], sh.red_mn[3]));
        const float gmx = fmaxf(fmaxf(sh.red_mx[0], sh.red_mx[1]), fmaxf(sh.red_mx[2], sh.red_mx[3]));
        dpl_octav_state z;
        z.sum = 0.0;
        z.cnt_gt = 0ull;
        z.cnt_le = 0ull;
        z.min_enc = gmn <= gmx ? enc_f32(gmn) : 0xFFFFFFFFu;
        z.max_enc = gmn <= gmx ? enc_f32(gmx) : 0u;
        z.nan_seen = sh.low_nan != 0u ? 1u : 0u;
        z.unsigned_div = ud;
        z.n_elems = n_pair;
        z.len[0] = 0u;
        z.len[1] = 0u;
        z.cur = 2u;
        z.reserved = 0u;
        z.s = s0;          // (a restart begins at s_0)
        z.iters = 0u;
        z.done = 0u;
        z.mode = 2u;
        if (route == 0u) {
            z.done = 1u;
        } else if (bad && rescued) {
            z.mode = 3u;   // restart from s_0 on the pair's exact bracket: its units go on the rescue's work list
            const uint32_t nu = (uint32_t)((n_pair + kRescueUnit - 1) / kRescueUnit);
            const uint32_t e = atomicAdd(&ctl->len[0], 1u), u0 = atomicAdd(&ctl->len[1], nu);
            fa.missed[3 * e] = pair;
            fa.missed[3 * e + 1] = u0;
            fa.missed[3 * e + 2] = nu;
        } else if (bad) {
            z.mode = 1u;   // restart from s_0 on the compaction route: state as k_octav_update<true> leaves it
            atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
        } else {
            z.s = s;
            z.iters = evals;
            z.done = 1u;
        }
        *me = z;
        atomicAdd(&ctl->sum, (double)L_listed);   // the batch's listed values (statistics)
        // history: the bin this pair asked for (whatever became of its walk)
        if (!small && route != 0u) atomicMax(fa.vis_w + (size_t)tensor * kLogWords, (uint32_t)kLogNB - sh.jwant);
    }
    DPL_PROF_T(wt5);
    DPL_PROF_ADD(0, wt0, wt1);
    DPL_PROF_FLUSH();
    DPL_PROF_ADD(3, wt4, wt5);
    if (tid == 0) {
        g_prof_iters_add(blockIdx.x, evals);
        DPL_PROF_L(L_listed);
    }
}

// One workgroup per pair (largest first): the pair's only HBM read, then its walk.
__global__ __launch_bounds__(kThreads, DPL_TAIL_OCC) void k_octav_tail(
    const dpl_work_item* __restrict__ slices, const float* const* __restrict__ segs, dpl_octav_state* __restrict__ st,
    uint32_t n_tensors, const uint64_t* __restrict__ pair_base, float* __restrict__ list0, dpl_octav_state* __restrict__ ctl,
    const dpl_span* __restrict__ spans, unsigned long long* __restrict__ rows, const TailArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* l_packed = reinterpret_cast<unsigned long long*>(lds_raw);
    __shared__ Shared sh;
    const uint32_t tid = threadIdx.x;
    DPL_PROF_T(kt0);
    const dpl_work_item it = slices[blockIdx.x];
    const bool part = it.reserved > 1u;    // a slice of a pair of several (uniform): streamed like a pair, its row left for k_octav_tail_merge
    const uint32_t pair = it.slot, cnt = it.count;
    const float* pg = segs[it.seg] + it.offset;
    const uint32_t tensor = pair % n_tensors;
    const bool small = !part && cnt <= kSmallCap;   // lists its whole window: every step exact
    // The pair's list region: pair_base[pair] .. pair_base[pair + 1] (list_cap_of(elements) values; a pair of c slices: c equal
    // parts, slice j lists into part j).
    const uint64_t region0 = pair_base[pair];
    const uint32_t region = (uint32_t)(pair_base[pair + 1] - region0);
    uint32_t cap = region;
    uint64_t list_at = region0;
    if (part) {
        const unsigned long long n = spans[pair].count, c = it.reserved;
        const unsigned long long per = (((n + c - 1ull) / c) + 3ull) & ~3ull;     // (dpl_build_octav_slices' cut)
        cap = (uint32_t)((region / (uint32_t)c) & ~31u);
        list_at = region0 + ((it.offset - spans[pair].offset) / per) * (unsigned long long)cap;
    }
    for (int b = tid; b < kLogNB + kWave; b += kThreads) l_packed[b] = 0ull;
    if (tid == 0) {
        sh.list_cap = cap;
        sh.region_cap = region;
        const uint32_t hist = small ? 0u : fa.pred[(size_t)tensor * kPredRow];   // kLogNB - bin; 0: none (a cold start lists from bin 1 and raises)
        sh.tail_j = (hist >= 1u && hist < (uint32_t)kLogNB) ? (uint32_t)kLogNB - hist : 1u;
        sh.cursor = 0u;
        sh.low_sum = 0.0;
        sh.low_cnt = 0u;
        sh.low_nan = 0u;
    }
    __syncthreads();
    stream_tail(pg, cnt, reinterpret_cast<uint32_t*>(list0 + list_at), sh, ctl, !small);
    __syncthreads();   // every LDS histogram add has landed; the per-wave ranges and the out-of-window sums are in sh
    DPL_PROF_T(kt1);
    if (part) {
        // the packed histogram row (16 KiB per 4 MiB read), the list's length and final threshold bin in its word 0; range,
        // out-of-window sums and NaN flag by atomics on the pair's freshly initialised state
        unsigned long long* row = rows + (size_t)blockIdx.x * kLogNB;
        for (int b = tid; b < kLogNB; b += kThreads)
            row[b] = b == 0 ? ((unsigned long long)sh.tail_j << 32) | (unsigned long long)sh.cursor : l_packed[b];
        if (tid == 0) {
            dpl_octav_state* me = st + pair;
            const float gmn = fminf(fminf(sh.red_mn[0], sh.red_mn[1]), fminf(sh.red_mn[2], sh.red_mn[3]));
            const float gmx = fmaxf(fmaxf(sh.red_mx[0], sh.red_mx[1]), fmaxf(sh.red_mx[2], sh.red_mx[3]));
            if (gmn <= gmx) {
                atomicMin(&me->min_enc, enc_f32(gmn));
                atomicMax(&me->max_enc, enc_f32(gmx));
            }
            if (sh.low_cnt) {
                atomicAdd(&me->sum, sh.low_sum);
                atomicAdd(reinterpret_cast<unsigned long long*>(&me->cnt_gt), (unsigned long long)sh.low_cnt);
            }
            if (sh.low_nan) atomicOr(&me->nan_seen, 1u);
        }
        return;
    }
#if defined(DPL_TAIL_ABL_NOWALK)
    if (tid == 0) {
        st[pair].done = 1u;
        st[pair].s = sh.red_mx[0];
    }
#if defined(DPL_TAIL_ABL_DELAY)   // a stand-in for the walk's latency: DPL_TAIL_ABL_DELAY ticks spent by every wave (1) or by wave 0 alone (2: the others exit)
    if (DPL_TAIL_ABL_WHO == 2 && tid >= kWave) return;
    {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)DPL_TAIL_ABL_DELAY) __builtin_amdgcn_s_sleep(8);
    }
#endif
    return;
#endif
    walk_tail<kTailVec>(pair, tensor, lds_raw, sh, st, ctl, pair_base, list0, fa, cnt);
    DPL_PROF_T(kt2);
    DPL_PROF_ADD(4, kt0, kt1);
    DPL_PROF_ADD(5, kt1, kt2);
}

// ---- pairs of more than one slice (> 1 044 480 elements: the packed histogram's 20-bit counts) -------------------------------
// A slice is streamed like a pair of its own (k_octav_tail above, `part`: histogram in LDS, its own threshold, its values listed
// into its part of the pair's list region) and leaves its packed histogram row (16 KiB per 4 MiB read), its list's length and final
// threshold bin (word 0 of the row) and — by atomics on the pair's freshly initialised state — range, out-of-window sums and
// NaN flag.  k_octav_tail_merge, one workgroup per such pair behind it: the rows added up in LDS (a bin that holds 2^20 values or
// more does not fit the packed word: such a pair goes to the compaction route), the slices' lists moved together, then the
// SAME walk (walk_tail) over the merged histogram and list.
__global__ __launch_bounds__(kThreads, DPL_TAIL_OCC) void k_octav_tail_merge(
    const dpl_work_item* __restrict__ slices, dpl_octav_state* __restrict__ st, uint32_t n_tensors,
    const uint64_t* __restrict__ pair_base, float* __restrict__ list0, dpl_octav_state* __restrict__ ctl,
    const dpl_span* __restrict__ spans, const unsigned long long* __restrict__ rows, const uint32_t* __restrict__ pair_order,
    const uint32_t* __restrict__ pair_slice0, const TailArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* l_packed = reinterpret_cast<unsigned long long*>(lds_raw);
    __shared__ Shared sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = pair_order[blockIdx.x];           // largest first: the pairs of more than one slice come first
    const uint32_t sl0 = pair_slice0[2 * pair], sl1 = pair_slice0[2 * pair + 1];
    const uint32_t tensor = pair % n_tensors;
    const unsigned long long n_pair = spans[pair].count;
    // ---- the rows added up (a thread: bins tid, tid + 256, ...), the counts watched
    constexpr int kPer = kLogNB / kThreads;
    unsigned long long acc[kPer];
    uint32_t over = 0u, nsum = 0u;
    double dsum = 0.0;          // (the window's total from the slices' own words: what s_0 is made of should the merged words not hold)
#pragma unroll
    for (int q = 0; q < kPer; ++q) acc[q] = 0ull;
    for (uint32_t sl = sl0; sl < sl1; ++sl) {
        const unsigned long long* row = rows + (size_t)sl * kLogNB;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int b = (int)tid + q * kThreads;
            const unsigned long long x = b == 0 ? 0ull : row[b];
            acc[q] += x;
            over |= (uint32_t)(acc[q] >> 63);       // (the count field is bits 43 .. 62: bit 63 set = 2^20 values or more in one bin)
            const uint32_t cx = (uint32_t)(x >> kPackShift);
            nsum += cx;
            dsum += bin_sum(x & kPackMask, cx, b);
        }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) l_packed[(int)tid + q * kThreads] = acc[q];
    for (int b = kLogNB + (int)tid; b < kLogNB + kWave; b += kThreads) l_packed[b] = 0ull;
    // ---- the slices' lists: lengths, thresholds, where each goes
    const uint32_t region = (uint32_t)(pair_base[pair + 1] - pair_base[pair]);
    const uint32_t part_cap = (region / (sl1 - sl0)) & ~31u;          // (as k_octav_tail cut the region)
    if (tid == 0) {
        uint32_t off = 0u, jmax = 1u, dropped = 0u;
        for (uint32_t sl = sl0; sl < sl1; ++sl) {
            const unsigned long long w0 = rows[(size_t)sl * kLogNB];
            const uint32_t len = (uint32_t)w0;
            dropped |= len > part_cap ? 1u : 0u;      // a slice that listed more than its part holds: nothing of it was kept
            sh.seg_off[sl - sl0] = off;
            sh.seg_len[sl - sl0] = len > part_cap ? 0u : len;
            off += len > part_cap ? 0u : len;
            jmax = max(jmax, (uint32_t)(w0 >> 32));
        }
        sh.list_cap = region;
        sh.region_cap = region;
        const dpl_octav_state* me = st + pair;
        sh.cursor = off;
        sh.tail_j = jmax;
        sh.low_sum = me->sum;
        sh.low_cnt = (uint32_t)me->cnt_gt;
        sh.low_nan = me->nan_seen;
        const bool any = me->min_enc <= me->max_enc;
        sh.red_mn[0] = any ? dec_f32(me->min_enc) : INFINITY;
        sh.red_mx[0] = any ? dec_f32(me->max_enc) : -INFINITY;
        for (int q = 1; q < kWaves; ++q) sh.red_mn[q] = INFINITY, sh.red_mx[q] = -INFINITY;
        sh.bad = dropped;        // (an incomplete list: the compaction route, like a bin of 2^20 values)
    }
    __syncthreads();
    if (__any(over != 0u) && (tid & (kWave - 1)) == 0) atomicOr(&sh.bad, 1u);   // (behind the barrier: thread 0 has set sh.bad above)
    __syncthreads();
    float* lp = list0 + pair_base[pair];
    // moved together towards the front, slice by slice, tile by tile: a tile is read whole before any of it is written (the
    // destination never lies behind the source, and never reaches a later tile's source)
    for (uint32_t sl = sl0 + 1; sl < sl1; ++sl) {
        const uint32_t len = sh.seg_len[sl - sl0], off = sh.seg_off[sl - sl0];
        const float* src = lp + (size_t)(sl - sl0) * part_cap;
        float* dst = lp + off;
        for (uint32_t i0 = 0; i0 < len; i0 += kThreads) {       // (uniform)
            const uint32_t i = i0 + tid;
            const float x = i < len ? __builtin_nontemporal_load(src + i) : 0.0f;
            __syncthreads();
            if (i < len) dst[i] = x;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // (no line of the region read above may answer the walk's loads from this CU's L1)
    __syncthreads();
    if (sh.bad) {   // a bin of 2^20 values or more: the compaction route (the state as walk_tail leaves it for that route)
        const double wd = wave_sum(dsum);
        const uint32_t wn = wave_sum(nsum);
        if ((tid & (kWave - 1)) == 0) {
            sh.red_d[tid / kWave] = wd;
            sh.red_a[tid / kWave] = wn;
        }
        __syncthreads();
        if (tid == 0) {
            dpl_octav_state* me = st + pair;
            dpl_octav_state z = *me;
            const float gmn = sh.red_mn[0], gmx = sh.red_mx[0];
            const bool nanseen = z.nan_seen != 0u;
            double tot = sh.low_sum;
            unsigned long long nz = sh.low_cnt;
            for (int q = 0; q < kWaves; ++q) tot += sh.red_d[q], nz += sh.red_a[q];
            // forward_net.py:324 — sum(|x|) / count(|x| > 0)
            const float s0 = nanseen ? __uint_as_float(0x7FC00000u) : __fdiv_rn((float)tot, (float)(long long)nz);
            z.sum = 0.0;
            z.cnt_gt = 0ull;
            z.cnt_le = 0ull;
            z.unsigned_div = (fa.dynamic_sym && fabsf(gmn) < 1e-6f && !nanseen) ? 4.0f : 1.0f;
            z.n_elems = n_pair;
            z.len[0] = 0u;
            z.len[1] = 0u;
            z.cur = 2u;
            z.reserved = 0u;
            z.s = s0;
            z.iters = 0u;
            z.done = (s0 != s0 || fa.max_iters <= 0) ? 1u : 0u;     // (NaN is a fixed point: finished)
            z.mode = z.done ? 2u : 1u;
            *me = z;
            if (!z.done) atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
            (void)gmx;
        }
        return;
    }
    walk_tail<kTailVec>(pair, tensor, lds_raw, sh, st, ctl, pair_base, list0, fa, (uint32_t)0u, n_pair);
}

// State + threshold snapshot of a batch: pred[t][0] = what the tensor's pairs asked for in the current and the previous epoch.
__global__ void k_octav_tail_init(dpl_octav_state* st, int64_t n_pairs, uint32_t* vis_w, const uint32_t* vis_o, uint32_t* pred,
                                  int64_t n_tensors, int zero_w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tensors) {
        const uint32_t mine = zero_w ? 0u : vis_w[i * kLogWords];
        if (zero_w) vis_w[i * kLogWords] = 0u;
        pred[i * kPredRow] = max(mine, vis_o[i * kLogWords]);
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
