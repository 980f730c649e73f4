// OCTAV ('-A mse', forward_net.py:284-342) in ONE read of the activations: the register-resident form.
//
// Why: measured on MI355X (scripts/mall_probe.hip, profiles/r02/mall_probe.txt) a second read of recently streamed
// data costs the same whether HBM or the 256 MiB Infinity Cache serves it (6.1-6.9 TB/s either way, and the two
// share one fabric), so a two-read form cannot pass ~40 % of the roofline.  The only storage that makes the second
// look free is on the CU: 512 KiB of VGPRs (+160 KiB LDS) per CU, 128 MiB per chip.
//
// How: persistent 256-thread workgroups (4 per CU, <= 128 VGPRs) pull SLICES (<= 25 600 elements of one
// (image, tensor) pair) from per-XCD queues and keep each slice in registers from its single HBM read until the
// pair no longer needs it:
//   single-slice pair   statistics, s_0 and every iteration run on the registers (count / sum of |x| > s per step).
//   multi-slice pair    (a cluster of workgroups, one slice each)
//     1  load + statistics + exact log-scale histogram of |x| in LDS (64 bins per octave: count and integer
//        mantissa sum per bin, as in the two-read bracket form);
//     2  merge the LDS histogram into the pair's row with agent-scope atomics, take a ticket;
//     3  the LAST arriver (leader) builds the suffix totals, runs the bracket walk over the bin edges and publishes
//        the bitmap of the bins the iterates can visit; the others poll the pair's flag;
//     4  every workgroup extracts ITS registers' values that fall in marked bins (about 2 %) into the pair's list,
//        takes a second ticket;
//     5  the last arriver loads the pair's list into its (now free) registers and walks the reference's exact
//        iteration: totals of the bins above the iterate's bin from the histogram (exact integers) + the listed
//        values of that bin (integer mantissa sums: the result does not depend on the order anything arrived in).
// Cross-workgroup traffic uses agent-scope atomics / sc1 stores and sc1 loads only (write-through, no L2
// write-back fences: MI355X_MICROARCH.md "inter-workgroup visibility").  Slices of a pair are adjacent in their
// queue and a workgroup only ever waits for slices queued before its own, so the cluster always completes, whatever
// the number of resident workgroups (a cluster is capped at kResMaxCluster slices).
// Pairs the bracket cannot serve (values >= 2^14 or inf, flat distributions, an iterate outside the marked bins, more
// listed values than a workgroup holds) finish on the compaction route of octav_kernels.hip, exactly as in the
// two-read form.
#include "common.hpp"
#include "octav_common.hpp"

#pragma clang fp contract(off)

namespace {

// keeps the scheduler from interleaving the unrolled per-vector bodies (their temporaries would not fit beside the
// resident slice)
#define DPL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// new SSA names for a resident vector at a phase boundary: nothing computed from it later (|x| bit patterns, keys) can be
// hoisted above this point and spilled across the wait
#define DPL_PIN4(q_) asm volatile("" : "+v"((q_).x), "+v"((q_).y), "+v"((q_).z), "+v"((q_).w))

constexpr int kResThreads = 256;
constexpr int kResWaves = kResThreads / kWave;
#ifndef DPL_RES_VEC
#define DPL_RES_VEC 20
#endif
constexpr int kResVec = DPL_RES_VEC;                              // 16-byte vectors per thread
constexpr uint32_t kResCap = (uint32_t)kResThreads * kResVec * 4; // elements a workgroup holds (25 600)
constexpr int kResQueueCap = 12;                                  // per-lane survivor queue; flushed above cap - 4
constexpr int kResQueueStride = kResQueueCap + 1;
constexpr int kResKeyWords = (1 << (31 - kLogShift)) / 32;        // bitmap over every 14-bit key: 512 words
constexpr int kResKeyWord0 = (int)(kLogKey0 >> 5);
constexpr uint32_t kResSpinLimit = 1u << 24;
constexpr uint32_t kResBigCluster = (1u << 20) / kResCap + 1;     // clusters this large may exceed the packed count field

// LDS: region A (16 KiB) = packed histogram -> S_ge (fp64) -> key bitmap + survivor queues; region B (8 KiB) = N_ge
constexpr int kResLdsA = kLogNB * 8;
constexpr int kResLdsB = kLogNB * 4;
static_assert(kResKeyWords * 4 + kResWaves * kWave * kResQueueStride * 4 <= kResLdsA, "queues must fit region A");

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

struct ResShared {
    double red_d[kResWaves];
    unsigned long long red_q[kResWaves];
    uint32_t red_a[kResWaves], red_b[kResWaves];
    float red_mn[kResWaves], red_mx[kResWaves];
    uint32_t bm[kLogWords + 2];
    uint32_t item, last, route;
    OctavStep step;
    int jb;
    uint32_t bad;
    double s_above;
    unsigned long long n_above;
};

// Pair row of the merged histogram -> raw per-bin (count, sum) in LDS -> suffix totals in place.
// The row is left zeroed for the next batch when `clean`.  All 256 threads.
__device__ __forceinline__ void res_load_suffix(unsigned long long* __restrict__ row, uint32_t* __restrict__ row_cnt,
                                                bool big, bool clean, uint32_t* n_ge, double* s_ge, ResShared& sh) {
    constexpr int kPerT = kLogNB / kResThreads;  // 8 consecutive bins per thread, thread 0 owns the TOP bins
    const int hi = kLogNB - 1 - (int)threadIdx.x * kPerT;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    uint32_t ln = 0;
    double ls = 0.0;
#pragma unroll
    for (int q = 0; q < kPerT; ++q) {
        const int b = hi - q;
        const unsigned long long v = ld_agent(row + b);
        uint32_t c;
        unsigned long long m;
        if (big) {
            c = ld_agent(row_cnt + b);
            m = v;
        } else {
            c = (uint32_t)(v >> kPackShift);
            m = v & kPackMask;
        }
        if (clean && v) st_agent(row + b, 0ull);
        if (clean && big && c) st_agent(row_cnt + b, 0u);
        const double sd = (double)(m + ((unsigned long long)c << 23)) * log_bin_scale(b);  // full 24-bit mantissas
        ln += c;
        ls += sd;
        n_ge[b] = c;
        s_ge[b] = sd;
    }
    // exclusive prefix over threads (thread order = descending bins): wave scan + serial pass over the wave totals
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
    for (int q = 0; q < kPerT; ++q) {  // own bins only: no other thread touches them
        const int b = hi - q;
        rn += n_ge[b];
        rs += s_ge[b];
        n_ge[b] = rn;
        s_ge[b] = rs;
    }
    __syncthreads();
}

__device__ __forceinline__ void res_route_compaction(dpl_octav_state* me, dpl_octav_state* ctl) {
    // state as k_octav_update<true> leaves it for the compaction route (s_0 and unsigned_div already stored)
    me->mode = 1u;
    me->iters = 0u;
    me->done = 0u;
    me->sum = 0.0;
    me->cnt_gt = 0ull;
    me->cnt_le = 0ull;
    me->len[0] = 0u;
    me->len[1] = 0u;
    me->cur = 2u;
    atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
}

#ifndef DPL_RES_OCC
#define DPL_RES_OCC 4
#endif
__global__ __launch_bounds__(kResThreads, DPL_RES_OCC) void k_octav_resident(
    const dpl_work_item* __restrict__ slices, const uint32_t* __restrict__ queue_begin, uint32_t* __restrict__ queue_head,
    int n_queues, const float* const* __restrict__ segs, dpl_octav_state* __restrict__ st, dpl_octav_state* __restrict__ ctl,
    uint32_t* __restrict__ sync, unsigned long long* __restrict__ lh, uint32_t* __restrict__ lh_cnt,
    uint32_t* __restrict__ bitmap, const uint64_t* __restrict__ pair_base, float* __restrict__ list0, int dynamic_sym,
    int max_iters, int fail_every) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* l_packed = reinterpret_cast<unsigned long long*>(lds_raw);
    double* s_ge = reinterpret_cast<double*>(lds_raw);
    uint32_t* keybm = reinterpret_cast<uint32_t*>(lds_raw);
    uint32_t* queues = reinterpret_cast<uint32_t*>(lds_raw) + kResKeyWords;
    uint32_t* n_ge = reinterpret_cast<uint32_t*>(lds_raw + kResLdsA);
    __shared__ ResShared sh;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const int w = tid / kWave;
    const int q = (int)(blockIdx.x % (uint32_t)n_queues);
    const uint32_t q0 = queue_begin[q], q1 = queue_begin[q + 1];

    for (;;) {
        // ------------------------------------------------------------------ next slice of this queue
        if (tid == 0) sh.item = q0 + add_agent(queue_head + q, 1u);
        __syncthreads();
        const uint32_t k = __builtin_amdgcn_readfirstlane(sh.item);   // scalar: the item and every pointer derived from it live in SGPRs
        if (k >= q1) break;
        const dpl_work_item it = slices[k];
        const uint32_t pair = it.slot, n_sl = it.reserved, cnt = it.count;
        dpl_octav_state* me = st + pair;
        const float* pg = segs[it.seg] + it.offset;
        const bool big = n_sl >= kResBigCluster;

        // ------------------------------------------------------------------ 1. the slice's only HBM read
        // [head: < 4 elements up to the first 16-byte boundary][nvec vectors][tail: < 4 elements].  The ragged ends
        // (at most 6 elements) ride in one extra register of threads 0..5; vectors past the end are +0.0 padding.
        f4 v[kResVec];
        const uint32_t head = min((uint32_t)(((16u - (uint32_t)((uintptr_t)pg & 15u)) & 15u) >> 2), cnt);
        const uint32_t nvec = (cnt - head) >> 2, n_rag = head + ((cnt - head) & 3u);
        // validity of vector u of this thread is  u * 256 < rem  (a constant against ONE loop-variant register: an index
        // per vector would be hoisted out of the slice loop as 25 loop invariants and spilled)
        const int rem = (int)nvec - (int)tid;
        {
            // buffer loads: the descriptor's byte count makes the hardware return +0.0 past the last vector — no branch,
            // no address clamp, and all kResVec loads of the thread are in flight at once (~100 KB per workgroup)
            // (one descriptor per vector row: the hardware's range check leaves the SGPR offset out, so the row offset
            // goes into the base and the byte count is what is left of the slice from there — scalar arithmetic only)
            const uint32_t voff = tid << 4;
            const int nbytes = (int)(nvec << 4);
#pragma unroll
            for (int u = 0; u < kResVec; ++u) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(pg + head + u * kResThreads * 4), 0, max(nbytes - u * kResThreads * 16, 0), 0x00020000);
                v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2 /* nt */));
            }
        }
        float vt = 0.0f;
        const bool has_rag = tid < n_rag;
        if (has_rag) vt = ((gptr_f32)pg)[tid < head ? tid : tid + (nvec << 2)];
        for (int b = tid; b < kLogNB; b += kResThreads) l_packed[b] = 0ull;
        __syncthreads();
        // padding is +0.0: never counted by the histogram, never above an iterate; only min / max must skip it
        float mn = INFINITY, mx = -INFINITY;
        uint32_t nan = 0u, nz = 0u;
        double sum = 0.0;
        auto stat1 = [&](float x) {
            mn = fminf(mn, x);
            mx = fmaxf(mx, x);
        };
        {
            auto hist1 = [&](float x) {
                // as LogHistOp (octav_kernels.hip): window bins 1 .. kLogNB-1 carry {count, 23 explicit mantissa bits};
                // nonzero values outside the window (rare) are accumulated directly
                const uint32_t bits = __float_as_uint(x);
                const uint32_t t = ((bits >> kLogShift) & 0x3FFFu) - (kLogKey0 + 1u);
                if (t < (uint32_t)(kLogNB - 1)) {
                    atomicAdd(l_packed + t + 1u, (1ull << kPackShift) | (unsigned long long)(bits & 0x7FFFFFu));
                } else if (__any(!(fabsf(x) <= 0.0f))) {
                    const float a = fabsf(x);
                    if (a > 0.0f) {
                        sum += (double)a;
                        ++nz;
                    }
                    nan |= (a != a);
                }
            };
#pragma unroll
            for (int u = 0; u < kResVec; ++u) {
                if (u * kResThreads < rem) {
                    stat1(v[u].x);
                    stat1(v[u].y);
                    stat1(v[u].z);
                    stat1(v[u].w);
                }
                hist1(v[u].x);
                hist1(v[u].y);
                hist1(v[u].z);
                hist1(v[u].w);
            }
            if (has_rag) stat1(vt);
            hist1(vt);
        }
        // workgroup totals of the directly accumulated statistics
        {
            const float wmn = wave_min(mn), wmx = wave_max(mx);
            const uint32_t wnz = wave_sum(nz);
            const double wsum = wave_sum(sum);
            const uint32_t wnan = __any(nan) ? 1u : 0u;
            if (lane == 0) {
                sh.red_mn[w] = wmn;
                sh.red_mx[w] = wmx;
                sh.red_a[w] = wnz;
                sh.red_b[w] = wnan;
                sh.red_d[w] = wsum;
            }
        }
        __syncthreads();
        float tmn = INFINITY, tmx = -INFINITY;
        uint32_t tnz = 0u, tnan = 0u;
        double tsum = 0.0;
#pragma unroll
        for (int j = 0; j < kResWaves; ++j) {
            tmn = fminf(tmn, sh.red_mn[j]);
            tmx = fmaxf(tmx, sh.red_mx[j]);
            tnz += sh.red_a[j];
            tnan |= sh.red_b[j];
            tsum += sh.red_d[j];
        }
        __syncthreads();

        // ------------------------------------------------------------------ 2. merge into the pair's row, ticket
        if (tid == 0) {
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
        }
        unsigned long long* row = lh + (uint64_t)pair * kLogNB;
        uint32_t* row_cnt = lh_cnt ? lh_cnt + (uint64_t)pair * kLogNB : nullptr;
        for (int b = tid; b < kLogNB; b += kResThreads) {
            const unsigned long long hv = l_packed[b];
            if (hv) {
                if (big) {
                    add_agent(row + b, hv & kPackMask);
                    add_agent(row_cnt + b, (uint32_t)(hv >> kPackShift));
                } else {
                    add_agent(row + b, hv);
                }
            }
        }
        drain_vmem();
        __syncthreads();
        uint32_t* sy = sync + (uint64_t)pair * 4u;
        if (tid == 0) sh.last = (add_agent(sy + 0, 1u) + 1u == n_sl) ? 1u : 0u;
        __syncthreads();
        const bool leader = __builtin_amdgcn_readfirstlane(sh.last) != 0u;
        uint32_t* brow = bitmap + (uint64_t)pair * kBitmapRow;
        uint32_t route;
        if (leader) {
            // -------------------------------------------------------------- 3. suffix totals, s_0, bracket walk
            res_load_suffix(row, row_cnt, big, false, n_ge, s_ge, sh);
            if (tid < (uint32_t)kLogWords) sh.bm[tid] = 0u;
            __syncthreads();
            if (tid == 0) {
                const double sum_out = __longlong_as_double((long long)ld_agent(reinterpret_cast<unsigned long long*>(&me->sum)));
                const BracketResult br = bracket_walk(
                    n_ge, s_ge, sh.bm, dec_f32(ld_agent(&me->min_enc)), dec_f32(ld_agent(&me->max_enc)),
                    ld_agent(&me->nan_seen) != 0u, sum_out, ld_agent(reinterpret_cast<unsigned long long*>(&me->cnt_gt)),
                    ld_agent(reinterpret_cast<unsigned long long*>(&me->n_elems)), dynamic_sym, max_iters);
                st_agent(reinterpret_cast<uint32_t*>(&me->s), __float_as_uint(br.s0));
                st_agent(reinterpret_cast<uint32_t*>(&me->unsigned_div), __float_as_uint(br.unsigned_div));
                st_agent(&me->iters, 0u);
                st_agent(&me->done, br.route == 0u ? 1u : 0u);
                st_agent(&me->mode, br.route == 1u ? 1u : 2u);
                st_agent(&me->len[0], 0u);
                st_agent(&me->len[1], 0u);
                st_agent(&me->cur, 2u);
                st_agent(reinterpret_cast<unsigned long long*>(&me->sum), 0ull);
                st_agent(reinterpret_cast<unsigned long long*>(&me->cnt_gt), 0ull);
                st_agent(reinterpret_cast<unsigned long long*>(&me->cnt_le), 0ull);
                if (br.route == 1u) atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->cnt_le), 1ull);
                sh.route = br.route;
                const int jmin = br.route == 2u ? br.jmin : kLogNB, jmax = br.route == 2u ? br.jmax : -1;
                sh.bm[kLogWords] = __float_as_uint(jmax < 0 ? INFINITY : log_edge(jmin));
                sh.bm[kLogWords + 1] = __float_as_uint(jmax < 0 ? -INFINITY : (jmax >= kLogNB - 1 ? INFINITY : log_edge(jmax + 1)));
            }
            __syncthreads();
            route = __builtin_amdgcn_readfirstlane(sh.route);
            if (tid < (uint32_t)kBitmapRow) st_agent(brow + tid, (route == 2u || tid >= (uint32_t)kLogWords) ? sh.bm[tid] : 0u);
            drain_vmem();
            __syncthreads();
            if (tid == 0) st_agent(sy + 1, 1u + route);   // flag: 1 + route
        } else {
            if (tid == 0) {
                uint32_t f = 0u, spins = 0u;
                while ((f = ld_agent(sy + 1)) == 0u && ++spins < kResSpinLimit) __builtin_amdgcn_s_sleep(8);
                if (f == 0u) atomicOr(&ctl->nan_seen, 2u);   // never expected: reported by dpl_octav_finalize
                sh.route = f ? f - 1u : 3u;
            }
            __syncthreads();
            route = __builtin_amdgcn_readfirstlane(sh.route);
            if (route == 2u && tid < (uint32_t)kBitmapRow) sh.bm[tid] = ld_agent(brow + tid);
        }
        __syncthreads();

        // ------------------------------------------------------------------ 4. this slice's values of the marked bins
        if (route == 2u) {
            for (int i = tid; i < kResKeyWords; i += kResThreads) {
                const int j = i - kResKeyWord0;
                keybm[i] = (j >= 0 && j < kLogWords) ? sh.bm[j] : 0u;
            }
            __syncthreads();
            uint32_t* myq = queues + (size_t)w * kWave * kResQueueStride + lane;   // entry j of lane l at [j][l]
#pragma unroll
            for (int u = 0; u < kResVec; ++u) DPL_PIN4(v[u]);
            asm volatile("" : "+v"(vt));
            uint32_t* dst = reinterpret_cast<uint32_t*>(list0 + pair_base[pair]);
            uint32_t qn = 0u;
            auto flush = [&]() {
                uint32_t inc = qn;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const uint32_t t = __shfl_up(inc, o, kWave);
                    if (lane >= (uint32_t)o) inc += t;
                }
                const uint32_t total = __shfl(inc, kWave - 1, kWave);
                uint32_t base = 0u;
                if (lane == kWave - 1) base = add_agent(&me->len[0], total);
                base = __shfl(base, kWave - 1, kWave) + inc - qn;
                for (uint32_t j = 0; j < qn; ++j) st_agent(dst + base + j, myq[j * kWave]);
                qn = 0u;
            };
            auto take4 = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
                const uint32_t wa = keybm[a >> (kLogShift + 5)], wb = keybm[b >> (kLogShift + 5)],
                               wc = keybm[c >> (kLogShift + 5)], wd = keybm[d >> (kLogShift + 5)];
                const uint32_t ha = (wa >> ((a >> kLogShift) & 31u)) & 1u, hb = (wb >> ((b >> kLogShift) & 31u)) & 1u,
                               hc = (wc >> ((c >> kLogShift) & 31u)) & 1u, hd = (wd >> ((d >> kLogShift) & 31u)) & 1u;
                if (__any((ha | hb | hc | hd) != 0u)) {   // branch-free append: the tail only advances for a survivor
                    myq[qn * kWave] = a;
                    qn += ha;
                    myq[qn * kWave] = b;
                    qn += hb;
                    myq[qn * kWave] = c;
                    qn += hc;
                    myq[qn * kWave] = d;
                    qn += hd;
                    if (__any(qn > (uint32_t)(kResQueueCap - 4))) flush();
                }
            };
#pragma unroll
            for (int u = 0; u < kResVec; ++u) {
                take4(__float_as_uint(v[u].x) & 0x7FFFFFFFu, __float_as_uint(v[u].y) & 0x7FFFFFFFu,
                      __float_as_uint(v[u].z) & 0x7FFFFFFFu, __float_as_uint(v[u].w) & 0x7FFFFFFFu);
                DPL_SCHED_FENCE();
            }
            take4(__float_as_uint(vt) & 0x7FFFFFFFu, 0u, 0u, 0u);
            if (__any(qn != 0u)) flush();
            drain_vmem();
        }
        __syncthreads();
        if (tid == 0) sh.last = (add_agent(sy + 2, 1u) + 1u == n_sl) ? 1u : 0u;
        __syncthreads();
        // NO `continue` anywhere in this loop: several back-edges make the compiler split it into nested loops, and a
        // `if (tid == 0) store; continue;` then sends lane 0 and the other lanes of a wave round different back-edges —
        // the wave runs s_barrier under partial exec masks and the workgroup falls apart (found the hard way).
        if (__builtin_amdgcn_readfirstlane(sh.last) != 0u) {

        // ------------------------------------------------------------------ 5. last of the cluster: the exact walk
        const uint32_t mode = __builtin_amdgcn_readfirstlane(ld_agent(&me->mode));
        const bool walk = mode == 2u && __builtin_amdgcn_readfirstlane(ld_agent(&me->done)) == 0u;
        // suffix totals again (the leader may have been another workgroup) and the row is handed back zeroed
        res_load_suffix(row, row_cnt, big, true, n_ge, s_ge, sh);
        if (walk) {
        if (tid < (uint32_t)kLogWords) sh.bm[tid] = ld_agent(brow + tid);
        const uint32_t L = __builtin_amdgcn_readfirstlane(ld_agent(&me->len[0]));
        const unsigned long long n_elems = ld_agent(reinterpret_cast<unsigned long long*>(&me->n_elems));
        const float ud = __uint_as_float(ld_agent(reinterpret_cast<uint32_t*>(&me->unsigned_div)));
        float s = __uint_as_float(ld_agent(reinterpret_cast<uint32_t*>(&me->s)));
        // the listed values (bit patterns of |x|) into the registers the slice no longer needs
        f4 rv[kResVec];
        {
            // buffer loads again (zero past the list's end); sc1: the values were written through by other XCDs
            const float* lp = list0 + pair_base[pair];
            const int nbytes = (int)(min(L, kResCap) << 2);
            const uint32_t voff = tid << 4;
#pragma unroll
            for (int u = 0; u < kResVec; ++u) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(lp + u * kResThreads * 4), 0, max(nbytes - u * kResThreads * 16, 0), 0x00020000);
                rv[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 16 /* sc1 */));
            }
        }
        const uint32_t Lvec = (L + 3u) >> 2;   // vectors of the list in use
        auto marked = [&](int j) { return j > 0 && j < kLogNB - 1 && ((sh.bm[j >> 5] >> (j & 31)) & 1u); };
        auto load_above = [&](int j) {   // thread 0: exact totals of the bins above bin j
            sh.n_above = (j + 1 < kLogNB) ? (unsigned long long)n_ge[j + 1] : 0ull;
            sh.s_above = (j + 1 < kLogNB) ? s_ge[j + 1] : 0.0;
        };
        __syncthreads();
        if (tid == 0) {
            const int j0 = log_bin(s);
            sh.jb = j0;
            sh.bad = (!marked(j0) || L > kResCap) ? 1u : 0u;
            if (fail_every > 0 && pair % (uint32_t)fail_every == 0u) sh.bad = 1u;   // test hook: the restart path
            if (!sh.bad) load_above(j0);
        }
        __syncthreads();
        uint32_t iters = 0u, done = 0u, bad = sh.bad;
        int jb = sh.jb;
        while (!done && !bad) {
            // values of bin jb above s: bit patterns in (bits(s), lower edge of bin jb + 1)
            const uint32_t lo = __float_as_uint(s), hi = ((uint32_t)(jb + 1) + kLogKey0) << kLogShift;
            uint32_t c = 0u, ms = 0u;
            auto in1 = [&](float f) {
                const uint32_t u = __float_as_uint(f);
                const bool in = u > lo && u < hi;
                c += (uint32_t)in;
                ms += in ? (u & 0x7FFFFFu) : 0u;
            };
#pragma unroll
            for (int u = 0; u < kResVec; ++u) {
                if ((uint32_t)u * kResThreads < Lvec) {   // uniform
                    in1(rv[u].x);
                    in1(rv[u].y);
                    in1(rv[u].z);
                    in1(rv[u].w);
                }
            }
            unsigned long long msum = (unsigned long long)ms;
            c = wave_sum(c);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) msum += __shfl_xor(msum, o, kWave);
            if (lane == 0) {
                sh.red_a[w] = c;
                sh.red_q[w] = msum;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long tc = 0ull, tm = 0ull;
                for (int j = 0; j < kResWaves; ++j) {
                    tc += sh.red_a[j];
                    tm += sh.red_q[j];
                }
                const unsigned long long tg = sh.n_above + tc;
                const double ts = sh.s_above + (double)(tm + (tc << 23)) * log_bin_scale(jb);
                const OctavStep qs = octav_step(ts, tg, n_elems - tg, ud, s, iters, max_iters);
                sh.step = qs;
                if (!qs.done) {
                    const int jn = log_bin(qs.s);
                    if (!marked(jn)) {
                        sh.bad = 1u;   // the bracket did not foresee this bin
                    } else if (jn != sh.jb) {
                        load_above(jn);
                        sh.jb = jn;
                    }
                }
            }
            __syncthreads();
            const OctavStep qs = sh.step;
            s = qs.s;
            iters = qs.iters;
            done = qs.done;
            bad = done ? 0u : sh.bad;
            jb = sh.jb;
            __syncthreads();
        }
        if (tid == 0) {
            if (bad) {
                res_route_compaction(me, ctl);   // restart from s_0 (still in me->s) on the compaction route
            } else {
                me->s = s;
                me->iters = iters;
                me->done = 1u;
            }
        }
        }   // walk
        }   // last of the cluster
        // one latch block that cannot be duplicated (convergent): keeps jump threading from giving the loop a second back-edge
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_octav_resident_init(dpl_octav_state* st, int64_t n_pairs, uint32_t* sync, uint32_t* queue_head, int n_queues) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_queues) queue_head[i] = 0u;
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
    if (i < n_pairs) {
        sync[4 * i + 0] = 0u;
        sync[4 * i + 1] = 0u;
        sync[4 * i + 2] = 0u;
        sync[4 * i + 3] = 0u;
    }
}

}  // namespace

extern int g_exact_fail_every;   // octav_kernels.hip (dpl_test_hook_exact_fail_every)
int dpl_octav_fallback_route(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                             const float* const* d_seg_ptrs, dpl_octav_state* d_states, int64_t n_pairs,
                             const dpl_span* d_pair_spans, const uint64_t* d_pair_base, const uint32_t* d_pair_order,
                             float* d_list0, float* d_list1, int dynamic_sym, int max_iters, hipStream_t st);

extern "C" {

uint32_t dpl_octav_slice_cap(void) { return kResCap; }
int dpl_octav_resident_occupancy(void) { return DPL_RES_OCC; }

int64_t dpl_build_octav_slices(const dpl_span* spans, int64_t n_spans, int n_queues, dpl_work_item* out, int64_t cap,
                               uint32_t* queue_begin) {
    if (!spans || n_spans < 0 || n_queues < 1 || n_queues > 64) return fail_msg("dpl_build_octav_slices: bad arguments");
    // pairs to queues: largest first, each to the queue with the least elements so far; a queue keeps that order
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_spans > 0 ? n_spans : 1));
    int* qof = (int*)malloc(sizeof(int) * (size_t)(n_spans > 0 ? n_spans : 1));
    if (!order || !qof) {
        free(order);
        free(qof);
        return fail_msg("dpl_build_octav_slices: out of memory");
    }
    for (int64_t i = 0; i < n_spans; ++i) order[i] = i;
    // stable sort by count, descending (n_spans is a few thousand: merge sort through qsort_r is overkill; insertion by
    // buckets of equal size is what the data looks like, but keep it simple and O(n log n))
    struct Cmp {
        static int f(const void* a, const void* b, void* ctx) {
            const dpl_span* sp = (const dpl_span*)ctx;
            const int64_t ia = *(const int64_t*)a, ib = *(const int64_t*)b;
            if (sp[ia].count != sp[ib].count) return sp[ia].count > sp[ib].count ? -1 : 1;
            return ia < ib ? -1 : (ia > ib ? 1 : 0);
        }
    };
    qsort_r(order, (size_t)n_spans, sizeof(int64_t), Cmp::f, (void*)spans);
    uint64_t load[64] = {0};
    int64_t n_total = 0;
    int64_t per_q[64] = {0};
    for (int64_t oi = 0; oi < n_spans; ++oi) {
        const dpl_span& sp = spans[order[oi]];
        const uint64_t c = sp.count == 0 ? 0 : (sp.count + kResCap - 1) / kResCap;
        if (c > 64) {
            free(order);
            free(qof);
            snprintf(g_err, sizeof(g_err), "dpl_build_octav_slices: a pair of %llu elements needs %llu slices (max 64)",
                     (unsigned long long)sp.count, (unsigned long long)c);
            return -3;
        }
        int best = 0;
        for (int qi = 1; qi < n_queues; ++qi)
            if (load[qi] < load[best]) best = qi;
        qof[order[oi]] = best;
        load[best] += sp.count;
        per_q[best] += (int64_t)c;
        n_total += (int64_t)c;
    }
    if (out && queue_begin && n_total <= cap) {
        int64_t pos[65];
        pos[0] = 0;
        for (int qi = 0; qi < n_queues; ++qi) pos[qi + 1] = pos[qi] + per_q[qi];
        for (int qi = 0; qi <= n_queues; ++qi) queue_begin[qi] = (uint32_t)pos[qi];
        for (int64_t oi = 0; oi < n_spans; ++oi) {
            const dpl_span& sp = spans[order[oi]];
            if (sp.count == 0) continue;
            const uint64_t c = (sp.count + kResCap - 1) / kResCap;
            // equal slices, cut on multiples of 4 elements so that an aligned pair yields aligned slices
            const uint64_t per = (((sp.count + c - 1) / c) + 3) & ~3ull;
            int64_t& p = pos[qof[order[oi]]];
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
    free(qof);
    return n_total;
}

int dpl_octav_run_resident(const dpl_work_item* d_slices, int64_t n_slices, const uint32_t* d_queue_begin, int n_queues,
                           uint32_t* d_queue_head, int n_workgroups, uint32_t* d_sync, uint64_t* d_lh, uint32_t* d_lh_cnt,
                           uint32_t* d_bitmap, const dpl_work_item* d_items, int64_t n_items,
                           const uint32_t* d_block_begin, int64_t n_blocks, const float* const* d_seg_ptrs,
                           dpl_octav_state* d_states, int64_t n_pairs, const dpl_span* d_pair_spans,
                           const uint64_t* d_pair_base, const uint32_t* d_pair_order, float* d_list0, float* d_list1,
                           int dynamic_sym, int max_iters, dpl_stream_t s) {
    if (n_slices <= 0 || n_pairs <= 0) return 0;
    if (n_queues < 1 || n_queues > 64 || n_workgroups < n_queues) return fail_msg("dpl_octav_run_resident: bad queue / workgroup counts");
    if (!d_lh || !d_lh_cnt || !d_sync || !d_queue_head || !d_bitmap) return fail_msg("dpl_octav_run_resident: null scratch buffer");
    if (int e = check_blocks("dpl_octav_run_resident", n_items, d_block_begin, n_blocks)) return e;
    hipStream_t st = (hipStream_t)s;
    dpl_octav_state* ctl = d_states + n_pairs;
    hipLaunchKernelGGL(k_octav_resident_init, dim3(grid_for(n_pairs + 1, 256)), dim3(256), 0, st, d_states, n_pairs, d_sync,
                       d_queue_head, n_queues);
    hipLaunchKernelGGL(k_octav_resident, dim3((unsigned)n_workgroups), dim3(kResThreads), (size_t)(kResLdsA + kResLdsB), st,
                       d_slices, d_queue_begin, d_queue_head, n_queues, d_seg_ptrs, d_states, ctl, d_sync,
                       reinterpret_cast<unsigned long long*>(d_lh), d_lh_cnt, d_bitmap, d_pair_base, d_list0, dynamic_sym,
                       max_iters, g_exact_fail_every);
    DPL_LAUNCH_CHECK("k_octav_resident");
    if (max_iters > 0)
        return dpl_octav_fallback_route(d_items, n_items, d_block_begin, n_blocks, d_seg_ptrs, d_states, n_pairs, d_pair_spans,
                                        d_pair_base, d_pair_order, d_list0, d_list1, dynamic_sym, max_iters, st);
    return 0;
}

}  // extern "C"
