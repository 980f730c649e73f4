/*
 * dipoorlet_hip.h — C ABI of the MI355X (gfx950) activation-calibration core.
 *
 * Drop-in boundary for Dipoorlet's calibration hot path.  The reference has no native layer: the
 * arithmetic below is what its Python does with numpy on host copies of every activation
 * (citations are into the reference tree, dipoorlet/...).  Each entry point names the reference
 * code it replaces.  All device work is enqueued on the caller's HIP stream, never synchronises,
 * never allocates; every buffer is caller-owned device memory unless marked HOST.
 *
 * Conventions
 *   - return value: 0 = ok, negative = error; dpl_last_error() gives a thread-local message.
 *   - "slot"  : index of an accumulator (one per calibrated tensor, or per (image,tensor) pair).
 *   - "seg"   : index into a device table of base pointers (one per live activation tensor).
 *   - "span"  : `count` consecutive fp32 elements at `offset` inside a segment, feeding one slot.
 *   - spans are cut into work items (one workgroup each) by dpl_build_work_items().
 *   - fp32 min/max accumulators are kept order-encoded in uint32 so integer atomics apply:
 *         enc(f) = bits(f) ^ (bits(f) >> 31 ? 0xFFFFFFFF : 0x80000000)
 */
#ifndef DIPOORLET_HIP_H
#define DIPOORLET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPL_ABI_VERSION 21 /* 20: dpl_fake_quant_pre (the producer's ReLU / Add + ReLU inside the Q/DQ kernel); 21: dpl_stream_* */
#define DPL_MAX_BINS 16384 /* LDS-privatised histogram: bins * 4 B per workgroup */

typedef void* dpl_stream_t; /* hipStream_t */

typedef struct dpl_span {
    uint64_t offset; /* elements from the segment base */
    uint64_t count;  /* elements */
    uint32_t seg;
    uint32_t slot;
} dpl_span;

typedef struct dpl_work_item {
    uint64_t offset;
    uint32_t count;
    uint32_t seg;
    uint32_t slot;
    uint32_t reserved;
} dpl_work_item;

/* Per-slot histogram range, derived on device from the running min/max (np.histogram semantics,
 * forward_net.py:266-268): first/last outer edges, fp32 step, reciprocal for the index estimate. */
typedef struct dpl_hist_range {
    float first, last, step, inv;
    uint32_t zero_bin; /* bin that |x| == 0 falls in */
    uint32_t status;   /* 0 ok, 1 range not finite (numpy raises ValueError), 2 too many bins for range */
    float dmax;        /* max(gmax, -gmin) before the degenerate +-0.5 expansion */
    uint32_t exact_div; /* 1: use the correctly rounded divide for the index estimate */
} dpl_hist_range;

/* Per-(image,tensor) OCTAV state (forward_net.py:315-340). */
typedef struct dpl_octav_state {
    double sum;       /* sum |x| (first pass) / sum_{|x|>s} |x| */
    uint64_t cnt_gt;  /* count(|x|>0) (first pass) / count(|x|>s) */
    uint64_t cnt_le;  /* count(|x|<=s) */
    uint32_t min_enc, max_enc;
    uint32_t nan_seen;
    uint32_t done;
    float s;
    float unsigned_div; /* 1 or 4 */
    uint32_t iters;
    uint32_t mode;      /* 0: every evaluation re-reads the full data; 1: tail lists (dpl_octav_run_compact);
                           2: log-histogram bracket (dpl_octav_run_bracket) / one-read walk; 3: one-read rescue */
    uint64_t n_elems;   /* elements of the pair (counted by the first pass) */
    uint32_t len[2];    /* lengths of the two tail lists */
    uint32_t cur;       /* list holding the values above the previous iterate: 0, 1, or 2 = none yet */
    uint32_t reserved;  /* compaction route: float bits of the iterate the current tail list was built at */
} dpl_octav_state;

int dpl_abi_version(void);
const char* dpl_last_error(void);
/* 0 when the current HIP device is a gfx950 part; fills name (HOST buffer) when non-null. */
int dpl_device_info(char* name, int name_cap, int* compute_units, uint64_t* hbm_bytes);
/* A HIP stream of a given priority on the current device (hipStreamNonBlocking), for work that runs BESIDE a caller's stream: the
 * OCTAV pipeline's rescue / result kernels.  priority: what hipDeviceGetStreamPriorityRange reports — *greatest (-1: high) ...
 * *least (1: low) on gfx950; values outside the range are clamped.  A LOW-priority stream lives on a hardware queue of its own class:
 * it neither takes the workgroup slots the caller's streaming kernel is waiting for (a high-priority one does) nor shares a queue
 * with the caller's stream (a normal-priority one may, depending on the order the process created its streams: the two then run one
 * after the other).  torch cannot create one (it clamps priorities above 0 to 0): wrap the handle in torch.cuda.ExternalStream. */
int dpl_stream_priority_range(int* least, int* greatest);
int dpl_stream_create(int priority, dpl_stream_t* out);
int dpl_stream_destroy(dpl_stream_t s);

/* HOST-only: cut spans into work items of at most `chunk_elems` (multiple of 1024) elements.
 * Returns the number of items (may exceed cap: call again with a larger buffer), <0 on error. */
int64_t dpl_build_work_items(const dpl_span* spans, int64_t n_spans, uint64_t chunk_elems,
                             dpl_work_item* out, int64_t cap);
/* HOST-only: split the concatenated element stream of `spans` into n_blocks contiguous, equal shares
 * (cuts 4 KiB-aligned inside a span).  Writes the items in stream order and block_begin[0..n_blocks]
 * (workgroup b owns items [block_begin[b], block_begin[b+1])).  Returns the item count (call with
 * out = NULL to size the buffer).  Few large, equal shares stream faster from HBM than many small items. */
int64_t dpl_build_balanced_items(const dpl_span* spans, int64_t n_spans, int64_t n_blocks, dpl_work_item* out,
                                 int64_t cap, uint32_t* block_begin);

/* ---- running min / max: replaces ort_outs[i].max()/.min() per tensor per image
 *      (forward_net.py:220-235) and np.min/np.max over the per-image lists (basic_algorithm.py:21). */
int dpl_minmax_init(uint32_t* d_min_enc, uint32_t* d_max_enc, uint32_t* d_nan, int64_t n_slots, dpl_stream_t s);
/* d_block_begin: device copy of block_begin (n_blocks + 1 entries), or NULL with n_blocks == n_items. */
int dpl_minmax_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, uint32_t* d_min_enc,
                          uint32_t* d_max_enc, uint32_t* d_nan, dpl_stream_t s);
/* decode to fp32; a slot that saw a NaN yields NaN for both (numpy max/min propagate NaN). */
int dpl_minmax_finalize(const uint32_t* d_min_enc, const uint32_t* d_max_enc, const uint32_t* d_nan,
                        int64_t n_slots, float* d_min, float* d_max, dpl_stream_t s);
/* inverse of finalize for merged (e.g. all-reduced) fp32 ranges. */
int dpl_minmax_encode(const float* d_min, const float* d_max, int64_t n_slots, uint32_t* d_min_enc,
                      uint32_t* d_max_enc, uint32_t* d_nan, dpl_stream_t s);

/* ---- |x| histogram: replaces np.histogram(np.abs(x), int(bins), (0, data_max)) per tensor per image
 *      and the np.stack(hist).sum(0) (forward_net.py:265-280, basic_algorithm.py:37-38).
 *      Counts are bit-exact with numpy; d_hist is uint64 [n_slots, bins], accumulated in place. */
int dpl_hist_prepare(const float* d_min, const float* d_max, int64_t n_slots, int bins,
                     dpl_hist_range* d_ranges, dpl_stream_t s);
int dpl_abs_hist_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                            int64_t n_blocks, const float* const* d_seg_ptrs, const dpl_hist_range* d_ranges,
                            int bins, uint64_t* d_hist, dpl_stream_t s);
/* ---- percentile clip: replaces the python loop of basic_algorithm.py:40-53. d_clip: fp32 [n_slots,2]. */
int dpl_hist_percentile(const uint64_t* d_hist, const float* d_min, const float* d_max, int64_t n_slots,
                        int bins, double threshold, float* d_clip, dpl_stream_t s);

/* ---- OCTAV ("mse"): replaces forward_net.py:315-330 per (image,tensor) pair (slot = pair).
 *      dpl_octav_run enqueues the first pass plus 20 (pass, update) rounds; converged pairs exit early. */
/* d_states holds n_pairs + 1 entries: the last one is a control block (count of pairs in full-pass mode). */
int dpl_octav_init(dpl_octav_state* d_states, int64_t n_pairs, int list_mode, dpl_stream_t s);
int dpl_octav_run(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                  const float* const* d_seg_ptrs, dpl_octav_state* d_states, int64_t n_pairs, int dynamic_sym,
                  int max_iters, dpl_stream_t s);
/* Same iterate sequence with tail compaction (init with list_mode = 1): the first evaluation reads the data
 * once and writes the values above s_0 to d_list0; then ONE launch walks all remaining iterations with a
 * persistent workgroup per pair over the shrinking tail lists (~2.5x smaller per step).
 * d_pair_spans[pair] = where the pair's data lives (seg, offset, count); d_pair_base[pair] = element offset of
 * the pair's region in both lists (a region holds the pair's element count); d_pair_order = pair indices,
 * largest first (launch order of the per-pair workgroups), may be NULL.  Pairs whose iterate decreases
 * (degenerate data) finish on the full data. */
int dpl_octav_run_compact(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, dpl_octav_state* d_states,
                          int64_t n_pairs, const dpl_span* d_pair_spans, const uint64_t* d_pair_base,
                          const uint32_t* d_pair_order, float* d_list0, float* d_list1, int dynamic_sym,
                          int max_iters, dpl_stream_t s);
/* Same iterate sequence in TWO reads of the data and no tail lists (init with list_mode = 2): pass 1 takes the
 * statistics and an exact log-scale histogram of |x| (64 bins per octave over 2^-18..2^14: per bin a count
 * and an integer sum of mantissas); a per-pair bracket walk over the bin edges marks the few dozen bins the
 * iterates can visit; pass 2 gathers only those elements (about 2 %) into d_list0; a per-pair kernel then runs the
 * exact iteration from (exact totals of the bins above) + (gathered elements of the current bin), verifying
 * that every iterate lands in a marked bin.  Pairs it cannot serve (bracket explodes on flat / degenerate
 * distributions, values >= 2^14, failed verification) are finished by the compaction route above.
 * d_lh_cnt: uint32 [n_pairs, 2048]; d_lh_sum: uint64 [n_pairs, 2048] (scratch: after the bracket walk they hold the
 * suffix totals N_ge[j] / S_ge[j] (fp64 bits) the exact walk reads); d_bitmap: uint32 [n_pairs, 66]
 * (64 words of marks + the gathered value range as two float bit patterns). */
int dpl_octav_run_bracket(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                          int64_t n_blocks, const float* const* d_seg_ptrs, dpl_octav_state* d_states,
                          int64_t n_pairs, const dpl_span* d_pair_spans, const uint64_t* d_pair_base,
                          const uint32_t* d_pair_order, float* d_list0, float* d_list1, uint32_t* d_lh_cnt,
                          uint64_t* d_lh_sum, uint32_t* d_bitmap, int dynamic_sym, int max_iters, dpl_stream_t s);
/* The reference's result in ONE read of the data and one launch per batch — the EXACT-TAIL form (csrc/octav_tail.hpp, included by
 * csrc/octav_tail_host.hip; replaces forward_net.py:323-330's 21 numpy passes):
 *   k_octav_tail — one workgroup per SLICE (at most dpl_octav_slice_cap() elements of one (image, tensor) pair; all but the largest
 *     pairs are one slice): the slice's only HBM read yields min / max, an exact log-scale histogram of |x| in LDS (64 bins per
 *     octave over 2^-18 .. 2^14: per bin a count and an integer mantissa sum) and the LIST of the values at or above a threshold
 *     bin (per tensor: the lowest bin its pairs asked for in the last two epochs of batches — a pair asks for the bin above which
 *     1/256 of it lies —, raised on the fly by a wave that lists more than its budget: ~0.5 - 1.5 % of a pair); then the same
 *     workgroup WALKS the pair: the early iterates as LOWER BOUNDS from the histogram's suffix totals, the late ones exactly from the
 *     list; accepted only if it ends on two exact evaluations;
 *   k_octav_tail_merge — a pair of more than one slice: its slices are streamed like pairs (packed histogram row per slice in d_lh),
 *     one workgroup adds the rows up, moves the lists together and runs the same walk;
 *   the RESCUE of a refused pair, on the device and without the host: the walk forms the pair's exact bracket from its histogram
 *     and puts the pair on a work list (d_missed); k_octav_rescue_gather re-reads THOSE PAIRS ALONE (many workgroups per pair) for
 *     the bracket's bins, k_octav_walk_rescue walks every iterate of the reference's sequence, verified; what even that cannot
 *     finish (a bracket that cannot be formed: flat distributions, values >= 2^14; a list beyond its region) ends on the
 *     compaction route (dpl_octav_oneread_compaction).  The rescue kernels are launched behind every batch and return at once
 *     when the control block (d_states[n_pairs]) lists nothing.
 * (Rounds 2 - 3 listed the bins ALL iterates were predicted to visit and evaluated every iterate exactly — k_octav_oneread,
 * k_octav_probe, k_octav_sort, k_octav_walk[_sorted]: superseded in round 4, removed in round 5.)
 *   dpl_build_octav_slices (HOST): cuts every span (= pair), largest first, into ceil(count / cap) equal slices (multiples
 *     of 4 elements); item.reserved = the pair's slice count; pair_slice0[2 slot], [2 slot + 1] = first / one-past-last
 *     slice of the pair in slot `slot` (spans carry slots 0 .. n_spans-1).  Returns the slice count (call with out = NULL
 *     to size), -3 when a pair needs more than 64 slices (use dpl_octav_run_bracket for such a set).
 * One HOST struct carries a batch's buffers (all pointers are device pointers; dpl_octav_plan_bind below fills it):
 *   dpl_octav_oneread_prepare = state initialisation (no dpl_octav_init call) + the tensors' threshold snapshot;
 *   dpl_octav_oneread_stream  = k_octav_tail (+ k_octav_tail_merge);
 *   dpl_octav_oneread_finish  = the rescue (+ dpl_octav_oneread_compaction when job.compaction_inline);
 *   dpl_octav_run_oneread = all of them on one stream.
 * stream and finish may run on different streams (the rescue of batch i beside the streaming kernel of batch i + 1) when the
 * caller orders finish(i) after stream(i) and gives concurrently live batches their own d_states / d_pred / d_lh / d_rescue_bm /
 * d_missed / d_resc; d_vis is shared (maxima are only ever raised), d_list0 may be shared by batches whose stream calls are on
 * one stream, d_list1 by batches whose finish calls are on one stream. */
typedef struct dpl_octav_oneread_job {
    /* the tensor set's decomposition (static per plan) */
    const dpl_work_item* d_slices;   /* dpl_build_octav_slices */
    int64_t n_slices;
    const uint32_t* d_pair_slice0;   /* [n_pairs, 2] */
    const dpl_span* d_pair_spans;    /* [n_pairs]: where each pair's data lives */
    const uint64_t* d_pair_base;     /* [n_pairs + 1]: element offset of the pair's region in d_list0 / d_list1; a region holds
                                        d_pair_base[p + 1] - d_pair_base[p] = dpl_octav_list_cap(elements) values (a pair of c slices:
                                        c x dpl_octav_list_cap(slice)), a multiple of 32 (a region must not share a 128-byte line with its
                                        neighbour: the streaming workgroup reads back the list it has just written); what a pair lists or
                                        its rescue gathers beyond that is dropped and the pair finishes on the compaction route */
    const uint32_t* d_pair_order;    /* [n_pairs]: pair indices, largest first (the pairs of more than one slice are its first n_multi entries) */
    int64_t n_pairs, n_tensors, n_small;   /* n_small: pairs of at most dpl_octav_small_pair() elements (they list their whole window) */
    int64_t n_multi;                 /* pairs of more than one slice */
    const dpl_work_item* d_items;    /* the balanced partition of the same pairs (compaction route), as for dpl_octav_run_compact */
    int64_t n_items;
    const uint32_t* d_block_begin;
    int64_t n_blocks;
    /* this batch */
    const float* const* d_seg_ptrs;
    dpl_octav_state* d_states;       /* [n_pairs + 1]; the last one is the control block: len[0] = pairs rescued, cnt_le = pairs left
                                        for the compaction route, sum = values listed */
    uint64_t* d_lh;                  /* [slices of multi-slice pairs, 2048]: a packed histogram row per such slice (may be NULL with n_multi == 0) */
    uint32_t* d_pred;                /* [n_tensors, 128]: the threshold snapshot (word 0 of a tensor's row) */
    float* d_list0;                  /* listed values, region per pair: written by dpl_octav_oneread_stream, dead once it has run */
    float* d_list1;                  /* the rescue's gathered values, same layout (dpl_octav_oneread_finish) */
    const uint64_t* d_pair_base_full;/* [n_pairs + 1]: whole-pair regions (element count rounded up to 32) ... */
    float* d_clist0;                 /* ... in the compaction route's two lists (dpl_octav_oneread_compaction): only needed once a batch */
    float* d_clist1;                 /*     reports unfinished pairs (d_states[n_pairs].cnt_le != 0), or up front with compaction_inline */
    uint32_t* d_rescue_bm;           /* [n_pairs, 64]: exact bracket of a rescued pair */
    uint32_t* d_missed;              /* [n_pairs, 3]: (pair, first unit, units) of the rescued pairs */
    uint64_t* d_resc;                /* [n_pairs, 3072]: suffix totals of a rescued pair (2048 fp64 sums, 2048 u32 counts): what its second walk starts from */
    /* carried across batches */
    uint32_t* d_vis;                 /* [2, n_tensors, 64] threshold history: two epoch accumulators (word 0 of a tensor's row: 2048 - the lowest
                                        bin its pairs asked for); walks raise d_vis[write_epoch], cleared first when reset_epoch != 0 */
    int32_t write_epoch, reset_epoch;
    int32_t dynamic_sym, max_iters;
    int32_t compaction_inline;       /* 1: dpl_octav_oneread_finish ends with dpl_octav_oneread_compaction; 0: the caller reads d_states[n_pairs].cnt_le
                                        when the batch is done and calls it only when that is non-zero */
    int32_t reserved;
} dpl_octav_oneread_job;
uint32_t dpl_octav_slice_cap(void);
uint32_t dpl_octav_list_cap(uint64_t n_elements); /* values the list region of a single-slice pair / of one slice of n elements holds */
uint32_t dpl_octav_small_pair(void); /* pairs of at most this many elements list their whole window */
int64_t dpl_build_octav_slices(const dpl_span* spans, int64_t n_spans, dpl_work_item* out, int64_t cap, uint32_t* pair_slice0);
int dpl_octav_oneread_prepare(const dpl_octav_oneread_job* job, dpl_stream_t s);
int dpl_octav_oneread_stream(const dpl_octav_oneread_job* job, dpl_stream_t s);
int dpl_octav_oneread_finish(const dpl_octav_oneread_job* job, dpl_stream_t s);
int dpl_octav_oneread_compaction(const dpl_octav_oneread_job* job, dpl_stream_t s);
int dpl_octav_run_oneread(const dpl_octav_oneread_job* job, dpl_stream_t s);

/* The exact-tail form WITHOUT knowing any buffer's size (what forward_net.py:315-340 needs from a binding): a HOST plan over the
 * (image, tensor) pairs of one tensor-set geometry reports the bytes of every part of the workspace, uploads its static tables
 * into caller memory and fills the job.  Minimal use, one batch at a time on one stream:
 *     p = dpl_octav_plan_create(spans, B * T, T, 1024);  dpl_octav_plan_sizes(p, &z);
 *     [hipMalloc z.tables_bytes, z.history_bytes (memset 0), z.state_bytes, z.rescue_bytes, 2 x z.list_bytes, z.fallback_bytes, z.result_bytes]
 *     dpl_octav_plan_upload(p, d_tables, s);
 *     per batch k: dpl_octav_plan_bind(p, d_tables, d_history, d_state, d_rescue, d_list0, d_list1, d_fallback, d_seg_ptrs, k, dyn, 20, &job);
 *                  dpl_octav_run_oneread(&job, s);  dpl_octav_finalize(job.d_states, job.n_pairs, d_result, s);
 * spans: one per pair, slot = image * n_tensors + tensor = its index.  NULL (dpl_last_error) when a pair needs more than 64
 * slices (dpl_octav_run_bracket serves such a set).  n_blocks: workgroups of the compaction route's partition (1024).
 * A pipeline that keeps several batches in flight gives each its own d_state / d_rescue, one d_list0 per stream that carries
 * dpl_octav_oneread_stream, one d_list1 per stream that carries dpl_octav_oneread_finish, and may pass d_fallback = NULL:
 * job.compaction_inline is then 0 and the caller binds a fallback block (z.fallback_bytes) only when it reads a non-zero
 * d_states[n_pairs].cnt_le, then calls dpl_octav_oneread_compaction + dpl_octav_finalize for that batch. */
typedef struct dpl_octav_plan dpl_octav_plan;
typedef struct dpl_octav_workspace_sizes {
    uint64_t tables_bytes;   /* the static decomposition (dpl_octav_plan_upload fills it) */
    uint64_t history_bytes;  /* what the tensors' pairs asked for in earlier batches: zero it once per calibration run, share it between the run's batches */
    uint64_t state_bytes;    /* per batch in flight: pair states + control block, the tensors' threshold snapshot */
    uint64_t rescue_bytes;   /* per batch in flight: brackets, work list and suffix totals of rescued pairs, histogram rows of multi-slice pairs */
    uint64_t list_bytes;     /* ONE list buffer; a job takes two (d_list0, d_list1) */
    uint64_t fallback_bytes; /* the compaction route's two whole-pair lists */
    uint64_t result_bytes;   /* fp32 [n_pairs, 3] of dpl_octav_finalize */
    int64_t n_pairs, n_slices, n_multi, n_small;
} dpl_octav_workspace_sizes;
dpl_octav_plan* dpl_octav_plan_create(const dpl_span* spans, int64_t n_spans, int64_t n_tensors, int64_t n_blocks);
void dpl_octav_plan_destroy(dpl_octav_plan* plan);
int dpl_octav_plan_sizes(const dpl_octav_plan* plan, dpl_octav_workspace_sizes* out);
int dpl_octav_plan_upload(const dpl_octav_plan* plan, void* d_tables, dpl_stream_t s);
/* call_index: batches bound before this one on the same d_history in this calibration run (0 for the first: the history's
 * two epoch accumulators alternate every 8 batches).  d_fallback may be NULL (see above). */
/* HOST: the compaction route's lists for just the pairs that need them.  h_states: a host copy of job.d_states (n_pairs + 1
 * entries) taken after dpl_octav_oneread_finish has run and the control block reported unfinished pairs (cnt_le != 0);
 * h_base_out [n_pairs + 1] receives whole-pair regions for those pairs (mode 1, not done), empty regions for the rest.  Returns
 * the elements ONE list takes: upload the table, allocate two lists of that many floats, point job.d_pair_base_full /
 * d_clist0 / d_clist1 at them and call dpl_octav_oneread_compaction (+ dpl_octav_finalize) — instead of binding the
 * whole-batch block of fallback_bytes. */
int64_t dpl_octav_fallback_layout(const dpl_octav_state* h_states, int64_t n_pairs, uint64_t* h_base_out);
int dpl_octav_plan_bind(const dpl_octav_plan* plan, void* d_tables, void* d_history, void* d_state, void* d_rescue, void* d_list0,
                        void* d_list1, void* d_fallback, const float* const* d_seg_ptrs, int64_t call_index, int dynamic_sym,
                        int max_iters, dpl_octav_oneread_job* job);
/* TEST HOOKS (0 = off; return the previous setting): _exact_ makes the exact walk of dpl_octav_run_bracket and the walk
 * of the exact-tail form reject every `every`-th pair, so that the restart paths — taken in production only when an iterate
 * leaves the gathered bins — can be exercised; _rescue_ does the same to the one-read form's rescue walk (-> compaction route). */
int dpl_test_hook_exact_fail_every(int every);
int dpl_test_hook_rescue_fail_every(int every);
/* d_out: fp32 [n_pairs,3] = (optimal_s, min, max) like the reference's per-image lists. */
int dpl_octav_finalize(const dpl_octav_state* d_states, int64_t n_pairs, float* d_out, dpl_stream_t s);

/* ---- per-output-channel weight ranges: replaces np.min/np.max(tensor.reshape(C,-1), -1)
 *      (basic_algorithm.py:88-90).  d_w row-major [rows, cols]. */
int dpl_rowwise_minmax(const float* d_w, int64_t rows, int64_t cols, float* d_min, float* d_max, dpl_stream_t s);

/* ---- fused QuantizeLinear->DequantizeLinear (quantize.py:197-239; ONNX opset-13 semantics) and the
 *      reference-owned torch restatement quant_acti (weight_transform/ada_quant_layer.py:28-36):
 *      y = (clamp(rint(x / scale[c]) + zp[c], qlo, qhi) - zp[c]) * scale[c]
 *      n_channels == 1: per tensor.  Otherwise channel c = (i / inner) % n_channels. */
int dpl_fake_quant(const float* d_x, float* d_y, int64_t n, const float* d_scale, const int32_t* d_zp,
                   int64_t n_channels, int64_t inner, int32_t qlo, int32_t qhi, dpl_stream_t s);
/* The same pair with its producer's activation applied on the way in, so that a forward that does not expose the producer's output
 * moves 8 B per element for ReLU -> Q/DQ instead of 16 (the reference's merge-ReLU rule leaves a ReLU behind Conv / Gemm / Add
 * unquantised at its input, quantize.py:50-55, so its OUTPUT carries the Q/DQ pair of the next layer, :74-93):
 *   DPL_FQ_PRE_NONE      y = fq(x)                        (d_x2 ignored)
 *   DPL_FQ_PRE_RELU      y = fq(max(x, 0))                NaN stays NaN, as np.maximum / torch.relu
 *   DPL_FQ_PRE_ADD_RELU  y = fq(max(x + x2, 0))           one fp32 addition (round to nearest); d_x2: n floats, no broadcast */
#define DPL_FQ_PRE_NONE 0
#define DPL_FQ_PRE_RELU 1
#define DPL_FQ_PRE_ADD_RELU 2
int dpl_fake_quant_pre(int32_t pre, const float* d_x, const float* d_x2, float* d_y, int64_t n, const float* d_scale,
                       const int32_t* d_zp, int64_t n_channels, int64_t inner, int32_t qlo, int32_t qhi, dpl_stream_t s);
/* The same arithmetic over a WHOLE tensor set in one launch (a caller that holds every tensor of a forward: one launch instead
 * of one per Q/DQ pair — 123 for ResNet-50, most of them launch-bound): d_items / d_block_begin = a partition of the tensors'
 * elements (dpl_build_balanced_items over one span per tensor: item.seg = tensor, offset / count in elements; items are cut on
 * multiples of 1024 elements inside a tensor), d_seg_x / d_seg_y = the tensors' input / output base pointers (may be equal:
 * in place), d_params[tensor] = its quantisation parameters (all DEVICE memory). */
typedef struct dpl_fake_quant_params {
    const float* d_scale;          /* [n_channels] */
    const int32_t* d_zero_point;   /* [n_channels] */
    int64_t n_channels;            /* 1: per tensor */
    int64_t inner;                 /* elements per channel row: channel c = (i / inner) % n_channels */
    int32_t qlo, qhi;
} dpl_fake_quant_params;
int dpl_fake_quant_items(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin, int64_t n_blocks,
                         const float* const* d_seg_x, float* const* d_seg_y, const dpl_fake_quant_params* d_params, dpl_stream_t s);

/* ---- cosine-similarity partial sums (utils.py:273-278): d_acc[slot*3 + {0,1,2}] += sum(a*b), sum(a*a),
 *      sum(b*b) in fp64. */
int dpl_cos_accumulate(const float* d_a, const float* d_b, int64_t n, double* d_acc, int64_t slot, dpl_stream_t s);

/* Bias correction (bias_correction.py:9-13: bias += mean(fp_out - q_out) over every axis but the channel one):
 * a, b viewed as [outer, n_channels, inner] (Conv output [n, C, H, W]; Gemm output [n, C] with inner = 1);
 * d_acc[c] += sum over outer and inner of (a - b), accumulated in fp64. */
int dpl_channel_diff_sum(const float* d_a, const float* d_b, int64_t outer, int64_t n_channels, int64_t inner,
                         double* d_acc, dpl_stream_t s);

/* The classifier head of a convolutional network in the calibration forward (forward_net.py:192-237 runs the graph with
 * onnxruntime; here: MIOpen + this): C[m, n] = alpha * sum_k A[m, k] * B(k, n) + beta * bias(m, n), fp32, sums over k in ascending
 * order with one fused multiply-add per term.  A: [m, k] row-major; B(k, n) = d_b[k * b_stride_k + n * b_stride_n] (an ONNX Gemm with
 * transB = 1 passes (1, k)); bias(m, n) = d_bias[m * bias_stride_m + n * bias_stride_n] or no bias (NULL).  Products of at most
 * DPL_GEMM_SMALL_MAX multiply-adds only (csrc/gemm_small.hip says why this exists and what it is not). */
#define DPL_GEMM_SMALL_MAX (1ull << 28)
/* A product with few column tiles of C is cut along k (the partial sums are added in ascending order; the number of cuts depends on
 * n and k only, so a row of C is the same sum whatever m is): d_workspace = dpl_gemm_small_workspace(m, n, k) bytes of device memory (0: none needed, NULL is fine). */
uint64_t dpl_gemm_small_workspace(int64_t m, int64_t n, int64_t k);
int dpl_gemm_small(const float* d_a, const float* d_b, const float* d_bias, float* d_c, int64_t m, int64_t n, int64_t k,
                   int64_t b_stride_k, int64_t b_stride_n, int64_t bias_stride_m, int64_t bias_stride_n, float alpha, float beta,
                   float* d_workspace, dpl_stream_t s);

/* Same sums per work-item slot (slot = (image, tensor) pair in the profiling flow, profiling.py:57-64):
 * d_acc[slot*3 + {0,1,2}] += sum(a*b), sum(a*a), sum(b*b); a from d_seg_a, b from d_seg_b (same geometry). */
int dpl_cos_items_accumulate(const dpl_work_item* d_items, int64_t n_items, const uint32_t* d_block_begin,
                             int64_t n_blocks, const float* const* d_seg_a, const float* const* d_seg_b,
                             double* d_acc, dpl_stream_t s);

/* ------------------------------------------------------------------ AdaRound / BRECQ / QDrop inner loop (N4)
 * Replaces the eager torch arithmetic of weight_transform/ada_quant_layer.py:28-50 (quant_acti, quant_weight),
 * :96-112 (adaround_reg), :115-116 (L2_norm), :147 (round-mask initialisation) and the torch.optim.Adam update of
 * adaround.py:119-133 / brecq.py:158-186.  Weights are viewed as [n_channels, inner] (output channel first; a
 * ConvTranspose weight is transposed by the caller as adaround.py:58-59 does); scale / q_min / q_max hold
 * n_channels entries (1 for per-tensor).  `clamp` follows the reference: only its per-channel branch clamps
 * (the per-tensor branch discards the clamp result, ada_quant_layer.py:44-45). */
typedef struct dpl_round_step_params {
    double lr, adam_beta1, adam_beta2, adam_eps; /* torch.optim.Adam defaults: 1e-3, 0.9, 0.999, 1e-8 */
    int32_t step;      /* Adam step t >= 1 of this update */
    int32_t adam;      /* 0: gradients only (mask, moments and weight are left untouched) */
    int32_t clamp;
    int32_t reserved;
    float grad_scale;  /* multiplies dL/d(qw): 1 / world_size after a SUM all-reduce (DDP's mean), else 1 */
    float reg_beta;    /* regulariser temperature of this iteration (TempDecay, ada_quant_layer.py:119-134); 0: off */
    float reg_lambda;  /* regulariser weight (adaround_reg.alpha = 0.01) */
    float reserved2;
} dpl_round_step_params;

/* Device-resident schedule of a learner, so that a whole iteration can be captured in a hipGraph and replayed:
 * dpl_round_sched_advance (one thread) sets this iteration's regulariser temperature (TempDecay over t_max
 * iterations) and Adam bias corrections, then counts the iteration; dpl_round_step reads them when d_sched != NULL
 * (p->step, p->reg_beta are then ignored).  Zero-initialise before the first iteration. */
typedef struct dpl_round_sched {
    int32_t iter;       /* iterations completed */
    int32_t adam_step;  /* Adam steps completed */
    float reg_beta, step_size, bc2_sqrt, reserved;
} dpl_round_sched;
int dpl_round_sched_advance(dpl_round_sched* d_sched, int32_t t_max, double lr, double adam_beta1, double adam_beta2,
                            dpl_stream_t s);

/* wfloor = floor(w / scale);  alpha = -log((zeta - gamma) / (w / scale - wfloor - gamma) - 1) */
int dpl_round_init(const float* d_w, const float* d_scale, int64_t n, int64_t n_channels, int64_t inner,
                   float* d_wfloor, float* d_alpha, dpl_stream_t s);
/* qw = clamp?(wfloor + h(alpha)) * scale;  h = rectified sigmoid (soft) or (alpha >= 0) (hard) */
int dpl_round_quant(const float* d_wfloor, const float* d_alpha, const float* d_scale, const float* d_qmin,
                    const float* d_qmax, int64_t n, int64_t n_channels, int64_t inner, int clamp, int soft,
                    float* d_qw, dpl_stream_t s);
/* One learning step in one pass: g = dL/d(alpha) from d_grad_qw (may be null) through the soft quantiser, plus
 * the regulariser's gradient; *d_reg_loss += lambda * sum(1 - |2h - 1|^beta) (may be null); Adam update of
 * alpha / m / v in place; d_qw_next (may be null) = the soft-quantised weight at the updated alpha;
 * d_grad_alpha (may be null) receives g. */
int dpl_round_step(const float* d_grad_qw, const float* d_wfloor, float* d_alpha, float* d_m, float* d_v,
                   const float* d_scale, const float* d_qmin, const float* d_qmax, int64_t n, int64_t n_channels,
                   int64_t inner, const dpl_round_step_params* p, const dpl_round_sched* d_sched, float* d_qw_next,
                   float* d_grad_alpha, double* d_reg_loss, dpl_stream_t s);
/* Sparse + quantised weight with a straight-through round (sparse_quant_layer.py:9-29, 61-66):
 * qw = clamp?(rint(w * mask / scale)) * scale; d_mask (0 / 1 per weight) may be null. */
int dpl_sparse_quant(const float* d_w, const float* d_mask, const float* d_scale, const float* d_qmin,
                     const float* d_qmax, int64_t n, int64_t n_channels, int64_t inner, int clamp, float* d_qw,
                     dpl_stream_t s);
/* Its gradient fused with torch.optim.SGD's update (sparse_quant.py:107-109: momentum, weight decay):
 * g = ((dL/dqw * grad_scale * scale) * clamp_pass) / scale * mask; d_grad_w (may be null) receives g; when `update`:
 * g += weight_decay * w; buf = first ? g : momentum * buf + g; w -= lr * buf. */
int dpl_sparse_step(const float* d_grad_qw, float* d_w, const float* d_mask, float* d_momentum_buf,
                    const float* d_scale, const float* d_qmin, const float* d_qmax, int64_t n, int64_t n_channels,
                    int64_t inner, int clamp, float grad_scale, float lr, float momentum, float weight_decay, int first,
                    int update, float* d_grad_w, dpl_stream_t s);
/* *d_loss += sum((y - target)^2) * inv_m with y = relu ? max(z, 0) : z  (L2_norm: inv_m = 1 / (elements / dim 1));
 * d_grad (may be null) = grad_coef * (y - target), zero where the ReLU is closed. */
int dpl_l2_loss(const float* d_z, const float* d_target, int64_t n, int relu, float grad_coef, double inv_m,
                float* d_grad, double* d_loss, dpl_stream_t s);
/* quant_acti with QDrop: y = rand < prob ? fake_quant(x) : x  (d_rand null: always quantised);
 * gradient as torch autograd defines it for the reference code: 0 through round(), 1 through the kept values. */
int dpl_acti_drop_fwd(const float* d_x, const float* d_rand, int64_t n, float scale, float qmin, float qmax,
                      float prob, float* d_y, dpl_stream_t s);
int dpl_acti_drop_bwd(const float* d_rand, const float* d_grad_y, int64_t n, float prob, float* d_grad_x,
                      dpl_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
