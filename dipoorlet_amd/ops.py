"""Torch-facing operators over the C ABI: device memory and streams come from PyTorch-ROCm, the
arithmetic runs in the hand-written gfx950 kernels (csrc/calib_kernels.hip).

Vocabulary: a *tensor set* is the list of activation tensors one forward of the network yields for a
batch of B calibration images (each tensor is [B, ...] contiguous fp32).  A `TensorSetPlan` cuts the
set into work items once; `CalibAccumulators` holds the persistent device-side statistics
(running min/max, uint64 histograms, OCTAV states) that replace the reference's per-image Python
lists (forward_net.py:204-235, 252-280, 297-340).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _hip

_SEG_CACHE_MAX = 64


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _require_cuda(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _hip.DipoorletHipError(f"{name} must be a ROCm device tensor; dipoorlet_amd has no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _hip.DipoorletHipError(f"{name} must be contiguous float32")


def _upload_struct_array(arr, n, device):
    """ctypes struct array -> device uint8 tensor (one-off, synchronous)."""
    nbytes = C.sizeof(arr._type_) * max(n, 1)
    host = torch.frombuffer(bytearray(C.string_at(C.addressof(arr), nbytes)), dtype=torch.uint8)
    return host.to(device)


# Workgroups per launch for the balanced partition (tuned on MI355X: few, large, equal shares stream
# faster from HBM than many small items; the histogram wants a little more latency hiding).
DEFAULT_BLOCKS = {"minmax": 256, "hist": 768, "octav": 1024, "cos": 512, "fq": 65536}   # measured optima per kernel family


def _blocks_for(kind):
    v = os.environ.get("DPL_BLOCKS_" + kind.upper())
    return int(v) if v else DEFAULT_BLOCKS[kind]


class WorkSet:
    """Device-resident work decomposition of one launch: items (+ block_begin for the balanced form)."""

    def __init__(self, items, n_items, block_begin, n_blocks):
        self.items, self.n_items, self.block_begin, self.n_blocks = items, n_items, block_begin, n_blocks

    def args(self):
        """(d_items, n_items, d_block_begin, n_blocks) as the C ABI takes them."""
        bb = _ptr(self.block_begin) if self.block_begin is not None else C.c_void_p(0)
        return _ptr(self.items), self.n_items, bb, self.n_blocks


class TensorSetPlan:
    """Static decomposition of a tensor set into work.

    elems_per_image[t] = elements of tensor t for ONE image; batch = images stacked along dim 0.
    Slots: tensor index t (per_image=False: images of a batch merge for free) or b * T + t (per image).
    Default form: the balanced partition (dpl_build_balanced_items) with a per-kernel workgroup count.
    chunk_elems forces the one-item-per-workgroup form (dpl_build_work_items) instead.
    """

    def __init__(self, elems_per_image, batch, device, chunk_elems=None):
        self.elems = [int(e) for e in elems_per_image]
        self.batch = int(batch)
        self.T = len(self.elems)
        self.device = torch.device(device)
        self.total = sum(self.elems) * self.batch
        self.chunk = int(chunk_elems) if chunk_elems else None
        self._work = {}
        self._seg_cache = {}

    @property
    def n_pairs(self):
        return self.batch * self.T

    def _spans(self, per_image):
        if per_image:
            return [(t, b * e, e, b * self.T + t) for b in range(self.batch) for t, e in enumerate(self.elems)]
        return [(t, 0, e * self.batch, t) for t, e in enumerate(self.elems)]

    def work(self, kind, per_image=False):
        """WorkSet for kernel family `kind` in {'minmax', 'hist', 'octav', 'cos', 'fq'}."""
        nb = None if self.chunk else max(1, min(_blocks_for(kind), (self.total + 4095) // 4096))
        key = (per_image, nb)
        w = self._work.get(key)
        if w is None:
            spans = self._spans(per_image)
            if self.chunk:
                arr, n = _hip.build_work_items(spans, self.chunk)
                w = WorkSet(_upload_struct_array(arr, n, self.device), n, None, n)
            else:
                arr, n, bb = _hip.build_balanced_items(spans, nb)
                bbt = torch.frombuffer(bytearray(bytes(bb)), dtype=torch.int32).to(self.device)
                w = WorkSet(_upload_struct_array(arr, n, self.device), n, bbt, nb)
            self._work[key] = w
        return w

    def octav_scratch(self):
        """(pair_spans, pair_base u64 [B*T], pair_order, list0, list1): where each pair's data lives, and two tail lists of
        the batch's size with the pair regions laid out in pair order (32-element aligned: whole 128-byte lines)."""
        if getattr(self, "_octav_scratch", None) is None:
            # (a region starts on a 128-byte line: the streaming workgroup that walks a pair reads back the list it has just
            # written, and no other workgroup of the launch may have pulled a line of it into the CU's L1 beforehand — which a
            # neighbouring pair's last, straddling line would)
            sizes = [((e + 31) // 32) * 32 for _ in range(self.batch) for e in self.elems]
            base = np.zeros(len(sizes) + 1, np.int64)       # (n + 1 entries: a region's size is the difference of two)
            base[1:] = np.cumsum(sizes)
            tot = int(sum(sizes))
            arr, ns = _hip._span_array(self._spans(True))
            order = np.argsort(-np.array(sizes, np.int64), kind="stable").astype(np.int32)  # largest pairs first
            self._octav_scratch = (_upload_struct_array(arr, ns, self.device), torch.from_numpy(base).to(self.device),
                                   torch.from_numpy(order).to(self.device),
                                   torch.empty(tot, dtype=torch.float32, device=self.device),
                                   torch.empty(tot, dtype=torch.float32, device=self.device))
        return self._octav_scratch

    def octav_loghist_scratch(self):
        """(count u32 [B*T, 2048], mantissa-sum u64 [B*T, 2048], bitmap u32 [B*T, 66]) for the bracket form."""
        if getattr(self, "_octav_lh", None) is None:
            n = self.n_pairs
            self._octav_lh = (torch.empty(n, 2048, dtype=torch.int32, device=self.device),
                              torch.empty(n, 2048, dtype=torch.int64, device=self.device),
                              torch.empty(n, 66, dtype=torch.int32, device=self.device))
        return self._octav_lh

    def octav_tail(self):
        """The exact-tail form's HOST plan (dpl_octav_plan_*: every buffer size comes from the C ABI), its static tables on the
        device and the threshold history of this tensor set — or None when a pair needs more than 64 slices."""
        if getattr(self, "_octav_tail", None) is None:
            arr, ns = _hip._span_array(self._spans(True))
            L = _hip.lib()
            handle = L.dpl_octav_plan_create(C.addressof(arr), ns, self.T, max(1, min(_blocks_for("octav"), (self.total + 4095) // 4096)))
            if not handle:
                self._octav_tail = False
            else:
                self._octav_tail = OctavTailPlan(self, handle)
        return self._octav_tail or None

    def octav_reset(self):
        """Cold start of the exact-tail OCTAV form: what the plan learned from earlier batches (the thresholds its tensors' pairs
        asked for) is forgotten — the state of a plan that has never run.  Call between independent calibration runs that share
        a plan (after OctavPipeline.sync())."""
        tp = getattr(self, "_octav_tail", None)
        if tp:
            tp.history.zero_()
            tp.calls = 0
        for pipe in list(getattr(self, "_octav_pipes", ())):
            pipe._forget(self)

    def _validate(self, tensors):
        if len(tensors) != self.T:
            raise _hip.DipoorletHipError(f"expected {self.T} tensors, got {len(tensors)}")
        for t, (x, e) in enumerate(zip(tensors, self.elems)):
            _require_cuda(x, f"tensor {t}")
            if x.numel() != e * self.batch:
                raise _hip.DipoorletHipError(f"tensor {t}: {x.numel()} elements, plan expects {e * self.batch}")

    def bind(self, tensors):
        """A tensor set a caller keeps RESIDENT and launches over again and again (pass 2 of `-A hist`, bench.py's pools):
        validated once, then held — the BoundSet owns references to the tensors, so their memory cannot be handed to another
        tensor while it lives, and a launch over it costs no per-tensor checks (557 tensors of a ViT-B/16 set: 0.1 - 0.2 ms of
        host time per launch otherwise, more than the host has to spare per batch)."""
        if isinstance(tensors, BoundSet):
            if tensors.plan is not self:
                raise _hip.DipoorletHipError("this tensor set is bound to another plan")
            return tensors
        tensors = list(tensors)
        self._validate(tensors)
        host = torch.tensor([x.data_ptr() for x in tensors], dtype=torch.int64).pin_memory()
        return BoundSet(self, tensors, host.to(self.device, non_blocking=True), host)

    def seg_table(self, tensors):
        """Device table of base pointers for this launch.  A list of tensors is checked on EVERY call (device, dtype, contiguity,
        element count: the allocator hands a freed set's addresses to other tensors, so a pointer seen before proves nothing);
        the table itself is cached per pointer tuple.  A BoundSet (bind()) was checked when it was bound.

        The tables live in ONE block allocated with the plan's first launch — _SEG_CACHE_MAX slots of T pointers on the device and
        the same in pinned host memory: a pointer tuple not seen before takes the next slot in rotation (written on the host,
        copied asynchronously on the current stream), so a launch over new tensors allocates nothing and does not wait for the
        device.  A slot is rewritten _SEG_CACHE_MAX distinct tensor sets later; no launch of this library is that far behind the
        host (the OCTAV pipeline runs at most DPL_OCTAV_PIPE_SETS batches ahead)."""
        if isinstance(tensors, BoundSet):
            if tensors.plan is not self:
                raise _hip.DipoorletHipError("this tensor set is bound to another plan")
            return tensors.table
        self._validate(tensors)
        key = tuple(x.data_ptr() for x in tensors)
        slot = self._seg_cache.get(key)
        if slot is not None:
            return self._seg_dev[slot]
        if getattr(self, "_seg_dev", None) is None:
            self._seg_host = torch.empty(_SEG_CACHE_MAX, max(self.T, 1), dtype=torch.int64).pin_memory()
            self._seg_np = self._seg_host.numpy()
            self._seg_dev = torch.empty(_SEG_CACHE_MAX, max(self.T, 1), dtype=torch.int64, device=self.device)
            self._seg_copied = [None] * _SEG_CACHE_MAX      # per slot: the event behind its last host -> device copy
            self._seg_keys = [None] * _SEG_CACHE_MAX
            self._seg_next = 0
        slot = self._seg_next % _SEG_CACHE_MAX
        self._seg_next += 1
        if self._seg_keys[slot] is not None:
            self._seg_cache.pop(self._seg_keys[slot], None)
        ev = self._seg_copied[slot]
        if ev is not None and not ev.query():       # (the copy that last read this pinned row has not run yet: 64 sets ago)
            ev.synchronize()
        self._seg_np[slot, :len(key)] = key
        self._seg_dev[slot].copy_(self._seg_host[slot], non_blocking=True)
        if ev is None:
            ev = self._seg_copied[slot] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._seg_keys[slot] = key
        self._seg_cache[key] = slot
        return self._seg_dev[slot]


class OctavTailPlan:
    """dpl_octav_plan of one TensorSetPlan: sizes, tables, history, and the buffers of the single-stream use (octav_batch)."""

    def __init__(self, plan, handle):
        self.plan, self.handle = plan, handle
        L = _hip.lib()
        self.sizes = _hip.OctavWorkspaceSizes()
        _hip.check(L.dpl_octav_plan_sizes(handle, C.byref(self.sizes)), "dpl_octav_plan_sizes")
        dev = plan.device
        self.tables = torch.empty(int(self.sizes.tables_bytes), dtype=torch.uint8, device=dev)
        # (no synchronisation here: the copies' sources are the C plan's own host tables, which live as long as this object —
        # __del__ waits for this event before it lets go of them)
        _hip.check(L.dpl_octav_plan_upload(handle, _ptr(self.tables), _stream()), "dpl_octav_plan_upload")
        self._uploaded = torch.cuda.Event()
        self._uploaded.record(torch.cuda.current_stream(dev))
        self.history = torch.zeros(int(self.sizes.history_bytes), dtype=torch.uint8, device=dev)
        self.calls = 0          # batches run on this history through octav_batch (pipelines count their own)
        self.n_multi = int(self.sizes.n_multi)
        self._single = None     # (state, rescue, list0, list1) of octav_batch
        self.fallback = None    # the compaction route's lists: allocated when a batch first reports unfinished pairs

    def __del__(self):
        try:
            self._uploaded.synchronize()
            _hip.lib().dpl_octav_plan_destroy(self.handle)
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def new(self, what):
        """A device buffer of the workspace part `what` ('state', 'rescue', 'list', 'fallback', 'result')."""
        n = int(getattr(self.sizes, what + "_bytes"))
        return torch.empty(max(n, 256), dtype=torch.uint8, device=self.plan.device)

    def compaction(self, job, states, stream, arena=None, host_states=None, base_host=None):
        """The compaction route for the pairs of `job`'s batch that neither their walk nor the rescue finished (its control block
        said so; rare — flat distributions, values beyond 2^14, lists beyond their regions, a dozen pairs of a cold first batch):
        dpl_octav_fallback_layout gives regions to just those pairs, and the route's two lists are that small (arena: a buffer to
        reuse; returned, grown if need be).
        host_states: the batch's states in PINNED host memory, already valid (the pipeline copies them behind every batch's rescue
        and reads them when the set comes up for reuse) — nothing here waits for the device then; without it the states are read
        back now (one host synchronisation on `stream`: octav_batch).  base_host: a pinned buffer of 8 (n_pairs + 1) bytes for the
        region table's upload (same rule)."""
        L = _hip.lib()
        n = self.plan.n_pairs
        if host_states is None:
            with torch.cuda.stream(stream):
                host_states = states[:(n + 1) * C.sizeof(_hip.OctavState)].cpu()
        if base_host is None:
            base = np.zeros(n + 1, np.uint64)
        else:
            base = base_host.numpy().view(np.uint64)[:n + 1]
        total = int(L.dpl_octav_fallback_layout(host_states.data_ptr(), n, base.ctypes.data))
        if total < 0:
            _hip.check(total, "dpl_octav_fallback_layout")
        if total == 0:
            return arena
        with torch.cuda.stream(stream):
            need = 2 * 4 * total + 8 * (n + 1)
            if arena is None or arena.numel() < need:
                arena = torch.empty(need + need // 2, dtype=torch.uint8, device=self.plan.device)
            tab = arena[2 * 4 * total:2 * 4 * total + 8 * (n + 1)]
            if base_host is None:
                tab.copy_(torch.from_numpy(base.view(np.uint8)))       # (a blocking copy of 8 (n + 1) bytes)
            else:
                tab.copy_(base_host[:8 * (n + 1)], non_blocking=True)
        job.d_pair_base_full = tab.data_ptr()
        job.d_clist0 = arena.data_ptr()
        job.d_clist1 = arena.data_ptr() + 4 * total
        _hip.check(L.dpl_octav_oneread_compaction(C.byref(job), C.c_void_p(stream.cuda_stream)), "dpl_octav_oneread_compaction")
        return arena

    def bind(self, state, rescue, list0, list1, tab, call_index, dyn, fallback=None):
        j = _hip.OctavOnereadJob()
        _hip.check(_hip.lib().dpl_octav_plan_bind(self.handle, _ptr(self.tables), _ptr(self.history), _ptr(state), _ptr(rescue),
                                                  _ptr(list0), _ptr(list1), _ptr(fallback) if fallback is not None else None,
                                                  _ptr(tab), int(call_index), dyn, _OCTAV_MAX_ITERS, C.byref(j)), "dpl_octav_plan_bind")
        return j

    def single(self):
        if self._single is None:
            self._single = (self.new("state"), self.new("rescue"), self.new("list"), self.new("list"))
        return self._single


class BoundSet:
    """A validated, resident tensor set of one plan (TensorSetPlan.bind): the tensors, kept alive, and their device pointer
    table.  Accepted wherever a list of the set's tensors is."""
    __slots__ = ("plan", "tensors", "table", "_host")

    def __init__(self, plan, tensors, table, host):
        self.plan, self.tensors, self.table, self._host = plan, tensors, table, host

    def __len__(self):
        return len(self.tensors)

    def __iter__(self):
        return iter(self.tensors)

    def __getitem__(self, i):
        return self.tensors[i]


class CalibAccumulators:
    """Persistent per-tensor statistics on the device."""

    def __init__(self, n_slots, device, bins=2048):
        self.n = int(n_slots)
        self.device = torch.device(device)
        self.bins = int(bins)
        if not (1 <= self.bins <= _hip.MAX_BINS):
            raise _hip.DipoorletHipError(f"bins must be in [1, {_hip.MAX_BINS}]")
        self.min_enc = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.max_enc = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.nan = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.gmin = torch.empty(self.n, dtype=torch.float32, device=self.device)
        self.gmax = torch.empty(self.n, dtype=torch.float32, device=self.device)
        self.hist = None
        self.ranges = None
        self.reset_minmax()

    def reset_minmax(self):
        _hip.check(_hip.lib().dpl_minmax_init(_ptr(self.min_enc), _ptr(self.max_enc), _ptr(self.nan), self.n,
                                              _stream()), "dpl_minmax_init")

    # ---- pass 1
    def minmax_accumulate(self, plan, tensors, per_image=False):
        """per_image=False: slot = tensor (n_slots = T).  per_image=True: slot = image * T + tensor
        (n_slots = B * T), the reference's one-entry-per-image lists."""
        tab = plan.seg_table(tensors)
        w = plan.work("minmax", per_image)
        if self.n < (plan.n_pairs if per_image else plan.T):
            raise _hip.DipoorletHipError("accumulator has fewer slots than the plan addresses")
        _hip.check(_hip.lib().dpl_minmax_accumulate(*w.args(), _ptr(tab), _ptr(self.min_enc), _ptr(self.max_enc),
                                                    _ptr(self.nan), _stream()), "dpl_minmax_accumulate")

    def finalize_minmax(self):
        """-> (gmin, gmax) fp32 device tensors [n_slots]."""
        _hip.check(_hip.lib().dpl_minmax_finalize(_ptr(self.min_enc), _ptr(self.max_enc), _ptr(self.nan), self.n,
                                                  _ptr(self.gmin), _ptr(self.gmax), _stream()),
                   "dpl_minmax_finalize")
        return self.gmin, self.gmax

    def set_minmax(self, gmin, gmax):
        """Install merged ranges (e.g. after an all-reduce across ranks)."""
        self.gmin.copy_(gmin)
        self.gmax.copy_(gmax)
        _hip.check(_hip.lib().dpl_minmax_encode(_ptr(self.gmin), _ptr(self.gmax), self.n, _ptr(self.min_enc),
                                                _ptr(self.max_enc), _ptr(self.nan), _stream()), "dpl_minmax_encode")

    # ---- pass 2
    def hist_prepare(self):
        """Derive per-tensor histogram ranges from gmin/gmax (call after finalize_minmax / set_minmax)."""
        if self.hist is None:
            self.hist = torch.zeros(self.n, self.bins, dtype=torch.int64, device=self.device)
            self.ranges = torch.empty(self.n * C.sizeof(_hip.HistRange), dtype=torch.uint8, device=self.device)
        else:
            self.hist.zero_()
        _hip.check(_hip.lib().dpl_hist_prepare(_ptr(self.gmin), _ptr(self.gmax), self.n, self.bins,
                                               _ptr(self.ranges), _stream()), "dpl_hist_prepare")

    def abs_hist_accumulate(self, plan, tensors):
        tab = plan.seg_table(tensors)
        w = plan.work("hist")
        _hip.check(_hip.lib().dpl_abs_hist_accumulate(*w.args(), _ptr(tab), _ptr(self.ranges), self.bins,
                                                      _ptr(self.hist), _stream()), "dpl_abs_hist_accumulate")

    def range_status(self):
        """HOST (synchronises): per-slot status from dpl_hist_prepare: 0 ok, 1 not finite, 2 too many bins."""
        raw = self.ranges.cpu().numpy().view(np.dtype([("first", "<f4"), ("last", "<f4"), ("step", "<f4"),
                                                       ("inv", "<f4"), ("zero_bin", "<u4"), ("status", "<u4"),
                                                       ("dmax", "<f4"), ("exact_div", "<u4")]))
        return raw

    def hist_percentile(self, threshold):
        clip = torch.empty(self.n, 2, dtype=torch.float32, device=self.device)
        _hip.check(_hip.lib().dpl_hist_percentile(_ptr(self.hist), _ptr(self.gmin), _ptr(self.gmax), self.n,
                                                  self.bins, float(threshold), _ptr(clip), _stream()),
                   "dpl_hist_percentile")
        return clip


_OCTAV_MAX_ITERS = 20  # forward_net.py:325


_OCTAV_MODE = {"full": 0, "compact": 1, "bracket": 2, "tail": 3}
_ONEREAD_EPOCH = 8      # batches per threshold-history epoch (the C ABI's: dpl_octav_plan_bind); a batch lists by what the
                        # pairs of the current and the previous epoch asked for (8 - 16 batches of history)


def _default_form():
    """DPL_OCTAV_FORM, else 'tail' (round 4: exact tail / bounded bulk; csrc/octav_tail.hpp)."""
    form = os.environ.get("DPL_OCTAV_FORM", "tail")
    if form not in _OCTAV_MODE:
        raise _hip.DipoorletHipError(f"DPL_OCTAV_FORM={form}: the forms are {sorted(_OCTAV_MODE)} (round 5 removed 'oneread')")
    return form


def octav_batch(plan, tensors, dynamic_sym, states=None, compact=None, form=None):
    """OCTAV for every (image, tensor) pair of one batch -> fp32 device tensor [B, T, 3] = (s, min, max).

    Forms (all end on the reference's result, forward_net.py:323-330):
      'tail' (default)     ONE read, one launch: statistics, exact log-scale histogram and the values at or above a threshold
                           bin (~1 % of a pair); the early iterates are taken as lower bounds from the histogram, the late
                           ones exactly from the list; a walk that does not end on two exact evaluations is rescued by the
                           exact two-read route (csrc/octav_tail.hpp).  Pairs of up to 64 slices (a pair above one slice —
                           dpl_octav_slice_cap() elements — is streamed slice by slice and walked by a merge kernel; a set with
                           a larger pair runs 'bracket'); every buffer is sized by the C ABI (dpl_octav_plan_*, OctavTailPlan).
                           This call reads the batch's control block back (one host synchronisation); OctavPipeline defers that
      'bracket'            two reads: statistics + exact log-scale histogram, bracket walk, gather of the marked
                           bins, exact per-pair iteration; pairs it cannot serve finish on the compaction route
      'compact'            evaluation at s_0 + tail compaction, then per-pair iteration over shrinking lists
      'full'               every evaluation re-reads the full data (21 passes)
    `compact=True/False` is the older spelling of 'compact' / 'full'.  DPL_OCTAV_FORM overrides the default.
    Tolerance: every form is within 1e-5 * max(1, |ref|) of the reference's optimal_s (the tests' bound; the reference's own stop
    rule is an absolute 1e-6); min / max are exact.  'tail' repeats to about 1e-6 relative from run to run, not bit for bit — the
    threshold history and wave timing decide which early iterates are taken as bounds, and with them the iterate on which
    |s' - s| < 1e-6 fires — while 'bracket' walks the reference's whole iterate sequence over exact integer sums and is bit-stable
    (the exact fall-back, and the reference point of the golden tests).
    states (optional): a uint8 device buffer of (B * T + 1) * 80 bytes that receives the pairs' states and the control block."""
    if form is None:
        form = ("compact" if compact else "full") if compact is not None else _default_form()
    mode = _OCTAV_MODE[form]
    if form == "tail":
        tp = plan.octav_tail()
        if tp is not None:
            return _octav_batch_tail(plan, tp, tensors, dynamic_sym, states)
        mode = 2            # a pair above 64 slices: the two-read form
    w = plan.work("octav", per_image=True)
    n_pairs = plan.n_pairs
    nbytes = (n_pairs + 1) * C.sizeof(_hip.OctavState)  # + control block
    if states is None or states.numel() < nbytes:
        states = torch.empty(nbytes, dtype=torch.uint8, device=plan.device)
    tab = plan.seg_table(tensors)
    L = _hip.lib()
    dyn = 1 if dynamic_sym else 0
    _hip.check(L.dpl_octav_init(_ptr(states), n_pairs, mode, _stream()), "dpl_octav_init")
    if mode == 0:
        _hip.check(L.dpl_octav_run(*w.args(), _ptr(tab), _ptr(states), n_pairs, dyn, _OCTAV_MAX_ITERS, _stream()),
                   "dpl_octav_run")
    else:
        spans, base, order, l0, l1 = plan.octav_scratch()
        if mode == 1:
            _hip.check(L.dpl_octav_run_compact(*w.args(), _ptr(tab), _ptr(states), n_pairs, _ptr(spans), _ptr(base),
                                               _ptr(order), _ptr(l0), _ptr(l1), dyn, _OCTAV_MAX_ITERS, _stream()),
                       "dpl_octav_run_compact")
        else:
            cnt, msum, bitmap = plan.octav_loghist_scratch()
            _hip.check(L.dpl_octav_run_bracket(*w.args(), _ptr(tab), _ptr(states), n_pairs, _ptr(spans), _ptr(base),
                                               _ptr(order), _ptr(l0), _ptr(l1), _ptr(cnt), _ptr(msum), _ptr(bitmap),
                                               dyn, _OCTAV_MAX_ITERS, _stream()), "dpl_octav_run_bracket")
    out = torch.empty(plan.batch, plan.T, 3, dtype=torch.float32, device=plan.device)
    _hip.check(L.dpl_octav_finalize(_ptr(states), n_pairs, _ptr(out), _stream()), "dpl_octav_finalize")
    return out


def _octav_batch_tail(plan, tp, tensors, dynamic_sym, states_out=None):
    """One batch of the exact-tail form on the caller's stream.  The compaction route (flat distributions, values beyond 2^14,
    lists beyond their regions: rare) needs lists for the pairs that take it, which exist from the first batch on that asks for
    them — so this reads the batch's control block back (the one host synchronisation of this call; OctavPipeline defers it)."""
    L = _hip.lib()
    tab = plan.seg_table(tensors)
    state, rescue, l0, l1 = tp.single()
    k = tp.calls
    tp.calls = k + 1
    job = tp.bind(state, rescue, l0, l1, tab, k, 1 if dynamic_sym else 0)
    job.compaction_inline = 0
    for fn in ("dpl_octav_oneread_prepare", "dpl_octav_oneread_stream", "dpl_octav_oneread_finish"):
        _hip.check(getattr(L, fn)(C.byref(job), _stream()), fn)
    csz = C.sizeof(_hip.OctavState)
    ctl = _hip.OctavState.from_buffer_copy(state[plan.n_pairs * csz:(plan.n_pairs + 1) * csz].cpu().numpy().tobytes())
    if ctl.cnt_le:
        tp.fallback = tp.compaction(job, state, torch.cuda.current_stream(plan.device), tp.fallback)
    out = torch.empty(plan.batch, plan.T, 3, dtype=torch.float32, device=plan.device)
    _hip.check(L.dpl_octav_finalize(_ptr(state), plan.n_pairs, _ptr(out), _stream()), "dpl_octav_finalize")
    if states_out is not None:      # (the caller's view of the pairs' states and the control block, as the other forms leave them)
        n = (plan.n_pairs + 1) * csz
        if states_out.numel() >= n:
            states_out[:n].copy_(state[:n])
    return out


_PIPE_SETS = max(2, int(os.environ.get("DPL_OCTAV_PIPE_SETS", "3")))   # batches the host may run ahead of the device (OctavPipeline)


_BESIDE = 0.5


def _behind(device, stream, others, cycles=1500000):
    """How far BEHIND work on `stream` does work on each of `others` run — i.e. do they share a hardware queue: 0 .. 1, the largest
    over `others`.  Asked of the device, in its own timestamps: a spin of under two milliseconds on `stream` between two timing
    events, a marker event on the other stream issued right after — a marker stamped at the spin's start ran beside it (separate
    queues: 0.01 - 0.03), one stamped at its end waited for it (one queue: 1.0).  The spin goes on the CANDIDATE and the marker on
    the other stream because `others` is usually the default stream: the other way round — the spin on the default stream of a
    process that has not used it for timing before — the marker was stamped at 0.5 - 0.6 of the spin beside it, and at 1.0 under
    rocprofv3 whatever the queues; this way round the readings are 0.01 or 1.0 with and without the profiler
    (scripts/stream_queue_probe.py).  HOST (synchronises; a few milliseconds, once per pipeline)."""
    worst = 0.0
    for o in others:
        s0, s1, m = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(stream):
            s0.record(stream)
            torch.cuda._sleep(cycles)
            s1.record(stream)
        m.record(o)
        s1.synchronize()
        m.synchronize()
        spin = max(s0.elapsed_time(s1), 1e-6)
        worst = max(worst, min(1.0, max(0.0, s0.elapsed_time(m) / spin)))
    return worst


def _runs_beside(device, stream, others):
    """Does work on `stream` run beside work on each of `others` (separate hardware queues)?"""
    return _behind(device, stream, others) < _BESIDE


def _separate_stream(device, others, tries=6):
    """A normal-priority stream whose work runs beside that of `others`: the first of `tries` pool streams that does — or, should
    none be clearly beside, the one that was least behind."""
    best, best_b = None, 2.0
    for _ in range(tries):
        s = torch.cuda.Stream(device)
        b = _behind(device, s, others)
        if b < best_b:
            best, best_b = s, b
        if b < _BESIDE:
            break
    return best


class OctavPipeline:
    """OCTAV over a RUN of batches in the exact-tail form on streams of its own.  Same kernels, same results as octav_batch.

        pipe = OctavPipeline(dynamic_sym)
        rows = [pipe.submit(plan, tensors) for ...]     # [B, T, 3] each, NOT valid yet
        pipe.sync()                                     # rows are valid for work on the caller's stream

    Lane streams (two, in rotation, each behind the caller's stream as of the submit): the streaming kernel, which also walks
    every pair — batch i + 1 starts while batch i drains.  Side stream, behind the streaming kernel of batch i and beside that
    of batch i + 1: the rescue of the pairs a walk could not finish (on the device, no host round trip), the result rows, the
    state and the threshold snapshot for batch i + 3.
    The pipeline OWNS its rotation per plan (every size is the C plan's, dpl_octav_plan_sizes: 2 S state blocks, S rescue blocks,
    one list per lane stream, one rescue list, the compaction route's lists once a batch has asked for them; the call counter):
    two pipelines may run the same plan (they share only what the plan has learned: maxima taken into its threshold history).
    The control block of a batch (listed values, rescued pairs, pairs left for the compaction route) is copied to pinned memory
    and read when the set comes up for reuse three submits later (or in sync()): statistics, and — only when the count is
    non-zero — the compaction route for that batch.  The activations of a batch, its pointer table and its result stay
    referenced from the set until then; the host runs at most three batches ahead (DPL_OCTAV_PIPE_SETS; with two the host waited
    for side-stream work that ends with the previous streaming kernel).  A tensor set with a pair above 64 slices runs
    octav_batch (the two-read form) on the caller's stream instead."""

    def __init__(self, dynamic_sym, device=None, lanes=None):
        self.dyn = 1 if dynamic_sym else 0
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        n_lanes = int(os.environ.get("DPL_OCTAV_LANES", "0")) or (2 if lanes is None else int(lanes))
        # The side stream: NORMAL priority, on a hardware queue of its own.  High priority (rounds 4 - 5) lets its kernels take over
        # from the streaming kernel's waves whenever the process' streams land on the queues that way (one stream 0.75 ms per batch
        # on the ResNet-50 sweep against 0.60; scripts/lanes1_after_lanes2.py, profiles/r06/ab_side_prio.txt); low priority (a stream
        # of the library's own making, dpl_stream_create) starves the rescue and with it the set the host waits for (0.77).  But a
        # normal-priority stream may SHARE its hardware queue with the caller's stream — four queues per priority class, handed out
        # in creation order — and then the rescue of batch i and the streaming kernel of batch i + 1 run one after the other (1.17 ms
        # under the profiler on one box, 0.71 on another after a two-lane pipeline had been created first): _separate_stream asks
        # the device.  DPL_OCTAV_SIDE_PRIO = -1 / 0 / 1 (high / normal / low): a fixed priority, no questions asked.
        self._side_handle = None
        prio = os.environ.get("DPL_OCTAV_SIDE_PRIO")
        if prio in (None, ""):
            self.side = _separate_stream(self.device, [torch.cuda.current_stream(self.device)])
        elif int(prio) > 0:
            h = C.c_void_p()
            with torch.cuda.device(self.device):
                _hip.check(_hip.lib().dpl_stream_create(int(prio), C.byref(h)), "dpl_stream_create")
            self._side_handle = h.value
            self.side = torch.cuda.ExternalStream(h.value, self.device)
        else:
            self.side = torch.cuda.Stream(self.device, priority=int(prio))
        # The streaming kernels of consecutive batches go to two streams of the pipeline's own in rotation (each behind the
        # caller's stream as of its submit), so that batch i + 1 starts while batch i drains: the last workgroups of a batch are
        # the pairs whose walks took longest (raised thresholds, long lists) and hold a few slots while the rest of the chip idles
        # — same-box A/B (scripts/mse_run.py): images alike +- 0, +- 10 % jitter - 3.5 %, feature maps at +- 30 % - 5.5 %,
        # ViT-B/16 - 0.7 %; three streams: + 3 ... 6 % (three kernels share the slots).  lanes = 1 (or DPL_OCTAV_LANES=1): the
        # caller's stream — what a caller that runs a network forward between two submits wants (forward_net_octav: beside the
        # next forward's convolutions the streaming kernel costs the forward 10 % and the loop 6 %, scripts/e2e_lanes_ab.sh).
        self.lanes = []
        if n_lanes >= 2:       # (each on a queue of its own too: two lanes on one queue would not overlap)
            for _ in range(n_lanes):
                self.lanes.append(_separate_stream(self.device, [torch.cuda.current_stream(self.device), self.side] + self.lanes))
        self._plans = {}          # id(plan) -> this pipeline's rotation state for the plan
        self._touched = []
        # record_events = True: per submit a (start, end) pair of timing events around the streaming kernel ON ITS LANE, kept in
        # .events (what --timing_json reports as the statistics' GPU seconds: the caller's stream no longer carries the kernel)
        self.record_events = False
        self.events = []
        # statistics: batches settled, batches / (image, tensor) pairs whose walk was refused (rescued or compaction route)
        self.reset_stats()

    def __del__(self):
        h = getattr(self, "_side_handle", None)
        if h:
            try:
                self.side.synchronize()
                _hip.lib().dpl_stream_destroy(C.c_void_p(h))
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass

    def reset_stats(self):
        self.batches = self.fallback_batches = self.fallback_pairs = self.compaction_pairs = 0
        self.tiles_reread = self.raises = 0
        self.list_share = self.max_share = 0.0    # listed values / elements (running mean / maximum over the settled batches)

    def _state(self, plan, tp):
        """This pipeline's rotation on `plan`; every buffer's size is the C plan's (dpl_octav_plan_sizes): 2 S state blocks (call k
        uses k % 2S; the block for call k + S is initialised at the end of call k's side-stream work, the one of call k must
        survive until the host has read k's control block, S submits later), S rescue blocks, one list per stream that carries
        streaming kernels, one rescue list (one side stream).  Nothing is shared with another pipeline on the same plan but what
        the plan has learned (the history's atomic maxima)."""
        ps = self._plans.get(id(plan))
        if ps is None:
            S = _PIPE_SETS
            csz = C.sizeof(_hip.OctavState)
            off = plan.n_pairs * csz
            states = [tp.new("state") for _ in range(2 * S)]
            ps = dict(plan=plan, tp=tp, calls=0, states=states, failed=[x[off:off + csz] for x in states],
                      list0=[tp.new("list") for _ in range(max(1, len(self.lanes)))], list1=tp.new("list"), fallback=None, sets=[])
            for _ in range(S):
                # (pinned: the control block — statistics —, the whole state block and the compaction route's region table: what
                # _settle needs of a batch is on the host by the time it looks, and what it uploads leaves without a wait)
                ps["sets"].append(dict(failed=torch.zeros(csz, dtype=torch.uint8).pin_memory(), rescue=tp.new("rescue"),
                                       host_states=torch.zeros(off + csz, dtype=torch.uint8).pin_memory(),
                                       base_host=torch.zeros(8 * (plan.n_pairs + 1), dtype=torch.uint8).pin_memory(),
                                       done=None, refs=None, pending=False, k=-1, prepared=None))
            self._plans[id(plan)] = ps
            if getattr(plan, "_octav_pipes", None) is None:
                import weakref
                plan._octav_pipes = weakref.WeakSet()
            plan._octav_pipes.add(self)
        return ps

    def scratch_bytes(self, plan):
        """Device bytes of this pipeline's OCTAV scratch on `plan`: tables, history, state / rescue blocks, lists, and the
        compaction route's lists if a batch has asked for them."""
        ps = self._plans.get(id(plan))
        if ps is None:
            return None
        tp = ps["tp"]
        n = tp.tables.numel() + tp.history.numel() + sum(x.numel() for x in ps["states"]) + sum(x.numel() for x in ps["list0"])
        n += ps["list1"].numel() + sum(st["rescue"].numel() for st in ps["sets"])
        return n + (ps["fallback"].numel() if ps["fallback"] is not None else 0)

    def _forget(self, plan):
        """plan.octav_reset(): this pipeline's rotation for the plan starts over (nothing may be in flight)."""
        ps = self._plans.get(id(plan))
        if ps is None:
            return
        for st in ps["sets"]:
            if st["pending"]:
                raise _hip.DipoorletHipError("octav_reset with batches in flight: call OctavPipeline.sync() first")
            st["prepared"] = None
            st["k"] = -1
        ps["calls"] = 0

    def _prepare(self, ps, st, k, stream):
        """State block + threshold snapshot for this pipeline's call number k on the plan (set `st` lends its rescue block to the
        job; prepare reads no tensors: the pointer table argument is a stand-in)."""
        r = k % (2 * _PIPE_SETS)
        job = ps["tp"].bind(ps["states"][r], st["rescue"], ps["list0"][0], ps["list1"], ps["list1"], k, self.dyn)
        _hip.check(_hip.lib().dpl_octav_oneread_prepare(C.byref(job), C.c_void_p(stream)), "dpl_octav_oneread_prepare")
        st["prepared"] = k

    def _finish(self, plan, ps, st):
        """Side stream: results of the set's batch -> its output rows, the set made ready for its next use, completion event."""
        side = self.side.cuda_stream
        _hip.check(_hip.lib().dpl_octav_finalize(_ptr(st["states"]), plan.n_pairs, _ptr(st["refs"][2]), side), "dpl_octav_finalize")
        self._prepare(ps, st, st["k"] + _PIPE_SETS, side)    # off the caller's stream: the set's next use is S calls away
        st["done"] = torch.cuda.Event()
        st["done"].record(self.side)

    def _settle(self, plan, ps, st):
        """HOST: read the statistics the set's last batch left in pinned memory (the walk finished long ago: the set comes up
        for reuse S submits later).  Nothing is launched here unless pairs are left for the compaction route: the pairs a walk
        could not finish are rescued on the device, without the host (submit)."""
        if not st["pending"]:
            return
        st["pending"] = False
        st["done"].synchronize()
        ctl = _hip.OctavState.from_buffer_copy(st["failed"].numpy().tobytes())
        # listed values; pairs rescued by a re-read of the pair + pairs that ended on the compaction route
        listed, failed = float(ctl.sum), int(ctl.len0) + int(ctl.cnt_le)
        self.compaction_pairs += int(ctl.cnt_le)
        self.raises += int(ctl.iters)              # thresholds raised on the fly (waves that listed beyond their budget)
        self.tiles_reread += int(ctl.reserved)     # 1024-element tiles holding a non-zero value outside the window
        self.batches += 1
        share = listed / max(1, plan.batch * sum(plan.elems))
        self.list_share = share if self.batches == 1 else 0.9 * self.list_share + 0.1 * share
        self.max_share = max(self.max_share, share)
        if failed:
            self.fallback_pairs += failed
            self.fallback_batches += 1
        if ctl.cnt_le:
            # what neither the walk nor the rescue could finish (a bracket that cannot be formed: flat distributions, values
            # beyond 2^14, a list beyond its region): the compaction route, launched only now that the count is known — the set's
            # batch is S submits old, its tensors are still referenced — on lists with regions for just the pairs that need them
            # (the states are read back), its results written over the batch's output rows
            ps["fallback"] = ps["tp"].compaction(st["job"], st["states"], self.side, ps["fallback"], st["host_states"], st["base_host"])
            self._finish(plan, ps, st)

    def submit(self, plan, tensors):
        form = _default_form()
        tp = plan.octav_tail() if form == "tail" else None
        if tp is None:      # another form was asked for, or a pair above 64 slices: one batch at a time on the caller's stream
            return octav_batch(plan, tensors, bool(self.dyn), form="bracket" if form == "tail" else form)
        main = torch.cuda.current_stream(plan.device)
        ps = self._state(plan, tp)
        sets = ps["sets"]
        k = ps["calls"]
        ps["calls"] = k + 1
        caller = main
        if self.lanes:
            main = self.lanes[k % len(self.lanes)]
        cur = sets[k % _PIPE_SETS]
        r = k % (2 * _PIPE_SETS)
        self._settle(plan, ps, cur)
        if cur["done"] is not None:
            main.wait_event(cur["done"])        # everything that last used this set has finished
        tab = plan.seg_table(tensors)
        out = torch.empty(plan.batch, plan.T, 3, dtype=torch.float32, device=plan.device)
        if main is not caller:
            main.wait_stream(caller)            # the batch's activations and its pointer table are the caller's stream's work
        cur["refs"] = (tensors if isinstance(tensors, BoundSet) else list(tensors), tab, out)
        cur["k"] = k
        cur["states"] = ps["states"][r]
        L = _hip.lib()
        if cur.get("prepared") != k:
            self._prepare(ps, cur, k, main.cuda_stream)
        job = cur["job"] = tp.bind(cur["states"], cur["rescue"], ps["list0"][k % len(ps["list0"])], ps["list1"], tab, k, self.dyn)
        if self.record_events:
            began = torch.cuda.Event(enable_timing=True)
            began.record(main)
        _hip.check(L.dpl_octav_oneread_stream(C.byref(job), C.c_void_p(main.cuda_stream)), "dpl_octav_oneread_stream")
        streamed = torch.cuda.Event(enable_timing=self.record_events)
        streamed.record(main)
        if self.record_events:
            self.events.append((began, streamed))
        self.side.wait_event(streamed)
        # the rescue of the pairs a walk could not finish, on the device: no host round trip decides anything
        _hip.check(L.dpl_octav_oneread_finish(C.byref(job), C.c_void_p(self.side.cuda_stream)), "dpl_octav_oneread_finish")
        with torch.cuda.stream(self.side):
            cur["failed"].copy_(ps["failed"][r], non_blocking=True)    # (statistics, and whether the compaction route is needed: _settle)
            # ... which lays its lists out from the pairs' states: 80 bytes per pair behind every batch (315 KB for a ResNet-50
            # batch) instead of a read-back — a host synchronisation with the side stream, 0.3 ms of an idle caller's stream —
            # when a batch asks for the route
            cur["host_states"].copy_(cur["states"][:cur["host_states"].numel()], non_blocking=True)
        self._finish(plan, ps, cur)
        cur["pending"] = True
        if all(p is not plan for p in self._touched):
            self._touched.append(plan)
        return out

    def sync(self):
        """Settle every outstanding batch (host waits for the walks), order the caller's stream after the side stream and
        let go of the batches' tensors."""
        for plan in self._touched:
            ps = self._plans[id(plan)]
            for st in sorted(ps["sets"], key=lambda q: q["k"]):
                self._settle(plan, ps, st)
        torch.cuda.current_stream(self.device).wait_stream(self.side)
        for lane in self.lanes:
            torch.cuda.current_stream(self.device).wait_stream(lane)
        for plan in self._touched:
            for st in self._plans[id(plan)]["sets"]:
                st["refs"] = None
        self._touched = []


# ------------------------------------------------------------------------------- single-tensor conveniences
def minmax(x):
    """(min, max) of one device tensor as a fp32 device tensor [2] (NaN if x holds a NaN)."""
    _require_cuda(x)
    plan = TensorSetPlan([x.numel()], 1, x.device)
    acc = CalibAccumulators(1, x.device)
    acc.minmax_accumulate(plan, [x])
    lo, hi = acc.finalize_minmax()
    return torch.stack([lo[0], hi[0]])


def abs_hist(x, bins, gmin, gmax):
    """np.histogram(|x|, bins, (0, max(gmax, -gmin)))[0] as an int64 device tensor [bins]."""
    _require_cuda(x)
    plan = TensorSetPlan([x.numel()], 1, x.device)
    acc = CalibAccumulators(1, x.device, bins)
    acc.set_minmax(torch.tensor([gmin], dtype=torch.float32, device=x.device),
                   torch.tensor([gmax], dtype=torch.float32, device=x.device))
    acc.hist_prepare()
    acc.abs_hist_accumulate(plan, [x])
    return acc.hist[0], acc


def rowwise_minmax(w2d, out=None):
    """Per-row (min, max) of a [rows, cols] fp32 device matrix (basic_algorithm.py:88-90); out = (lo, hi): where to write them."""
    _require_cuda(w2d, "w2d")
    rows, cols = w2d.shape
    if out is None:
        lo = torch.empty(rows, dtype=torch.float32, device=w2d.device)
        hi = torch.empty(rows, dtype=torch.float32, device=w2d.device)
    else:       # (a graph's initializers: slices of one result buffer, read back in one transfer)
        lo, hi = out
        _require_cuda(lo, "out[0]")
        _require_cuda(hi, "out[1]")
        if lo.numel() != rows or hi.numel() != rows:
            raise _hip.DipoorletHipError(f"rowwise_minmax: out holds {lo.numel()} / {hi.numel()} values for {rows} rows")
    _hip.check(_hip.lib().dpl_rowwise_minmax(_ptr(w2d), rows, cols, _ptr(lo), _ptr(hi), _stream()),
               "dpl_rowwise_minmax")
    return lo, hi


FQ_PRE = {None: 0, "none": 0, "relu": 1, "add_relu": 2}     # include/dipoorlet_hip.h DPL_FQ_PRE_*


def fake_quant(x, scale, zero_point, qlo, qhi, axis=None, out=None, pre=None, x2=None):
    """Fused QuantizeLinear -> DequantizeLinear (quantize.py:197-239) on the device.

    scale: fp32 device tensor [1] or [C]; zero_point: int32 device tensor of the same length;
    axis: channel axis when len(scale) > 1.  y = (clamp(rint(x/scale)+zp, qlo, qhi) - zp) * scale.
    pre: the producer's activation applied on the way in (dpl_fake_quant_pre) — 'relu': fq(max(x, 0)); 'add_relu':
    fq(max(x + x2, 0)), x2 of x's shape (the residual Add of a bottleneck and its ReLU) — for a forward that does not expose the
    producer's output (the merge-ReLU rule, quantize.py:50-55, puts the Q/DQ pair directly behind that ReLU).
    """
    _require_cuda(x, "x")
    code = FQ_PRE[pre]
    if code == 2:
        _require_cuda(x2, "x2")
        if x2.shape != x.shape or x2.dtype != torch.float32 or not x2.is_contiguous():
            raise _hip.DipoorletHipError("fake_quant(pre='add_relu'): x2 must be a contiguous fp32 tensor of x's shape")
    # (a graph walk issues one of these per tensor: no conversion calls when the parameters already are what the kernel takes)
    if not (scale.device == x.device and scale.dtype == torch.float32 and scale.dim() == 1 and scale.is_contiguous()):
        scale = scale.to(device=x.device, dtype=torch.float32).contiguous().reshape(-1)
    if not (zero_point.device == x.device and zero_point.dtype == torch.int32 and zero_point.dim() == 1 and zero_point.is_contiguous()):
        zero_point = zero_point.to(device=x.device, dtype=torch.int32).contiguous().reshape(-1)
    nch = scale.numel()
    if zero_point.numel() != nch:
        raise _hip.DipoorletHipError("scale and zero_point lengths differ")
    inner = 1
    if nch > 1:
        if axis is None:
            raise _hip.DipoorletHipError("per-channel fake_quant needs an axis")
        if x.shape[axis] != nch:
            raise _hip.DipoorletHipError(f"axis {axis} has {x.shape[axis]} channels, scale has {nch}")
        for d in x.shape[axis + 1:]:
            inner *= int(d)
    y = torch.empty_like(x) if out is None else out
    _hip.check(_hip.lib().dpl_fake_quant_pre(code, _ptr(x), _ptr(x2) if code == 2 else None, _ptr(y), x.numel(), _ptr(scale),
                                             _ptr(zero_point), nch, inner, int(qlo), int(qhi), _stream()), "dpl_fake_quant_pre")
    return y


class FakeQuantSet:
    """Fused QuantizeLinear -> DequantizeLinear over a WHOLE tensor set in one launch (dpl_fake_quant_items): for a caller that
    holds every tensor of a forward (the profiling flow's fp-vs-quantised comparison, quant_acti over a set) — one launch
    instead of one per Q/DQ pair, most of which are launch-bound.

        fq = FakeQuantSet(plan, params)          # params[t] = (scale fp32 [1 | C], zero_point int32 [1 | C], inner, qlo, qhi)
        ys = fq(xs)                              # or fq(xs, out=ys); xs[t]: [B, ...] contiguous fp32 as the plan describes

    `inner` = elements per channel row of ONE tensor as laid out in memory ([B, C, H, W] with per-channel parameters on axis 1:
    inner = H * W); ignored for per-tensor parameters."""

    def __init__(self, plan, params):
        if len(params) != plan.T:
            raise _hip.DipoorletHipError(f"expected {plan.T} parameter rows, got {len(params)}")
        self.plan = plan
        self.keep = []
        rows = (_hip.FakeQuantParams * plan.T)()
        for t, (scale, zp, inner, qlo, qhi) in enumerate(params):
            scale = scale.to(device=plan.device, dtype=torch.float32).contiguous().reshape(-1)
            zp = zp.to(device=plan.device, dtype=torch.int32).contiguous().reshape(-1)
            if scale.numel() != zp.numel():
                raise _hip.DipoorletHipError("scale and zero_point lengths differ")
            self.keep += [scale, zp]
            rows[t] = _hip.FakeQuantParams(scale.data_ptr(), zp.data_ptr(), scale.numel(), int(inner) if scale.numel() > 1 else 1,
                                           int(qlo), int(qhi))
        self.d_params = _upload_struct_array(rows, plan.T, plan.device)
        # the balanced partition over the batch's tensors (slot = tensor): 65536 workgroups of ~52 KB on the ResNet-50 set, so
        # that the resident ones read and write a dense window as the dispatcher hands them out (scripts/fq_set_blocks.py: with
        # 1024 resident workgroups of 3.3 MB each the read + write stream reaches 0.73 of 8 TB/s on some boxes of the pool and
        # 0.60 - 0.62 on others; 65536: 0.70 - 0.73 on both kinds)
        self.work = plan.work("fq")

    def __call__(self, tensors, out=None):
        plan = self.plan
        tx = plan.seg_table(tensors)
        if out is None:
            out = [torch.empty_like(x) for x in tensors]
        ty = plan.seg_table(out)
        _hip.check(_hip.lib().dpl_fake_quant_items(*self.work.args(), _ptr(tx), _ptr(ty), _ptr(self.d_params), _stream()),
                   "dpl_fake_quant_items")
        return out


def cos_accumulate(a, b, acc, slot=0):
    """acc[slot] += (sum a*b, sum a*a, sum b*b) in fp64 (utils.py:273-278 partial sums)."""
    _require_cuda(a, "a")
    _require_cuda(b, "b")
    if a.numel() != b.numel():
        raise _hip.DipoorletHipError("cos_accumulate: size mismatch")
    _hip.check(_hip.lib().dpl_cos_accumulate(_ptr(a), _ptr(b), a.numel(), _ptr(acc), slot, _stream()),
               "dpl_cos_accumulate")
    return acc


def channel_diff_sum(a, b, acc=None):
    """acc[c] += sum over every axis but the channel one of (a - b), fp64 (bias_correction.py:9-13).
    a, b: [n, C, spatial...] (channel axis 1) or [n, C]; returns the fp64 device tensor [C]."""
    _require_cuda(a, "a")
    _require_cuda(b, "b")
    if a.shape != b.shape or a.dim() < 2:
        raise _hip.DipoorletHipError("channel_diff_sum: shapes must match and carry a channel axis")
    n_ch = int(a.shape[1])
    inner = 1
    for d in a.shape[2:]:
        inner *= int(d)
    if acc is None:
        acc = torch.zeros(n_ch, dtype=torch.float64, device=a.device)
    _hip.check(_hip.lib().dpl_channel_diff_sum(_ptr(a), _ptr(b), int(a.shape[0]), n_ch, inner, _ptr(acc), _stream()),
               "dpl_channel_diff_sum")
    return acc


GEMM_SMALL_MAX = 1 << 28      # DPL_GEMM_SMALL_MAX (include/dipoorlet_hip.h)


def gemm_small(a, b, bias=None, alpha=1.0, beta=1.0):
    """alpha * a @ b + beta * bias through dpl_gemm_small (csrc/gemm_small.hip: the classifier head of a convolutional network,
    so that a calibration run of one never initialises torch's BLAS path).  a: [M, K] fp32 contiguous; b: [K, N] fp32, any strides (a transposed
    view of an [N, K] weight is what an ONNX Gemm with transB = 1 gives); bias: None, [N], [1, N], [M, 1] or [M, N].
    M * N * K <= GEMM_SMALL_MAX."""
    for t, name in ((a, "a"), (b, "b")) + (((bias, "bias"),) if bias is not None else ()):
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise _hip.DipoorletHipError(f"gemm_small: {name} must be a ROCm device tensor; dipoorlet_amd has no CPU path")
        if t.dtype != torch.float32:
            raise _hip.DipoorletHipError(f"gemm_small: {name} must be float32")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0]:
        raise _hip.DipoorletHipError("gemm_small: a [M, K] and b [K, N]")
    a = a.contiguous()
    M, K, N = int(a.shape[0]), int(a.shape[1]), int(b.shape[1])
    bp, sm, sn = None, 0, 0
    if bias is not None:
        bias = bias.expand(M, N)       # (broadcast axes get stride 0; raises if it does not broadcast)
        bp, sm, sn = _ptr(bias), int(bias.stride(0)), int(bias.stride(1))
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    L = _hip.lib()
    need = int(L.dpl_gemm_small_workspace(M, N, K))
    ws = torch.empty(need // 4, dtype=torch.float32, device=a.device) if need else None
    _hip.check(L.dpl_gemm_small(_ptr(a), _ptr(b), bp, _ptr(out), M, N, K, int(b.stride(0)), int(b.stride(1)), sm, sn,
                                float(alpha), float(beta), _ptr(ws) if need else None, _stream()), "dpl_gemm_small")
    return out


def cos_per_image(plan, tensors_a, tensors_b):
    """Cosine partial sums for every (image, tensor) pair of two tensor sets with the same geometry ->
    fp64 device tensor [B, T, 3] = (sum a*b, sum a*a, sum b*b)."""
    w = plan.work("cos", per_image=True)
    ta = plan.seg_table(tensors_a)
    tb = plan.seg_table(tensors_b)
    acc = torch.zeros(plan.batch, plan.T, 3, dtype=torch.float64, device=plan.device)
    _hip.check(_hip.lib().dpl_cos_items_accumulate(*w.args(), _ptr(ta), _ptr(tb), _ptr(acc), _stream()),
               "dpl_cos_items_accumulate")
    return acc
