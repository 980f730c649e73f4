"""`torch.ops.dipoorlet.*` — the custom-op spelling of the kernel library, for callers that live in
torch (ORT-IOBinding / eager pipelines, torch.compile graphs).  Registrations over dipoorlet_amd.ops: every op runs
the HIP kernels on the current stream and is registered for the 'cuda' (ROCm) device only — there is no CPU
implementation to fall back to.

Per tensor:
    dipoorlet::minmax(Tensor x) -> Tensor                      [2] fp32 (min, max); NaN if x holds one
    dipoorlet::abs_hist_(Tensor x, float dmax, int bins, Tensor(a!) hist) -> ()      hist += np.histogram(|x|, bins, (0, dmax))
    dipoorlet::hist_percentile(Tensor hist, float gmin, float gmax, float threshold) -> Tensor   [2] fp32 clip
    dipoorlet::octav(Tensor x, bool dynamic_sym) -> Tensor     [3] fp32 (s, min, max) (forward_net.py:315-330)
    dipoorlet::rowwise_minmax(Tensor w2d) -> (Tensor, Tensor)
    dipoorlet::fake_quant(Tensor x, Tensor scale, Tensor zero_point, int axis, int qlo, int qhi) -> Tensor
    dipoorlet::fake_quant_relu(Tensor x, Tensor scale, Tensor zero_point, int axis, int qlo, int qhi) -> Tensor          fq(relu(x))
    dipoorlet::fake_quant_add_relu(Tensor x, Tensor x2, Tensor scale, Tensor zero_point, int axis, int qlo, int qhi) -> Tensor
Over every tensor of a batch of images in ONE launch — what the reference's loops over `ort_outputs` stand for
(forward_net.py:220-235, 265-280, 314-340); xs[t] is tensor t of the batch, [B, ...] contiguous fp32:
    dipoorlet::minmax_batched(Tensor[] xs, Tensor(a!) mins, Tensor(b!) maxs) -> ()   running min / max per tensor, [T] fp32
    dipoorlet::abs_hist_batched_(Tensor[] xs, Tensor mins, Tensor maxs, int bins, Tensor(a!) hist) -> ()
                                                               hist[t] += np.histogram(|xs[t]|, bins, (0, max(|mins[t]|, |maxs[t]|)))
    dipoorlet::octav_batched(Tensor[] xs, bool dynamic_sym) -> Tensor                 [B, T, 3] fp32 (s, min, max) per (image, tensor)
    dipoorlet::fake_quant_set(Tensor[] xs, Tensor[] scales, Tensor[] zero_points, int[] inner, int[] qlo, int[] qhi) -> Tensor[]

No allocation inside except the outputs, no host synchronisation: everything a launch needs besides its inputs — the work
decomposition of the tensor set, the pointer-table slots, accumulators, OCTAV workspace — belongs to a plan cached on
(per-image sizes, batch, device) and is built by the FIRST call with that key; later calls only launch.  The OCTAV ops run the
exact-tail form through an ops.OctavPipeline on the caller's stream: the control block of a batch (how many pairs were rescued;
whether a pair is left for the rare compaction route) is read from pinned memory a few calls later, never waited for — the result
rows are ordered behind the kernels on the device (a stream wait).
"""
import ctypes
from typing import List, Tuple

import torch

from . import _hip, ops

_PLANS = {}
_PLANS_MAX = 32
_RANGE = {}          # device index -> fp32 [2]: the (gmin, gmax) argument of hist_percentile


class _Cached:
    """Everything the ops need for one (per-image sizes, batch, device) besides inputs and outputs."""

    def __init__(self, elems, batch, device):
        self.plan = ops.TensorSetPlan(elems, batch, device)
        self.device = device
        self.acc = {}        # bins -> CalibAccumulators (min / max encodings, histogram ranges; hist: the caller's tensor)
        self.pipes = {}      # dynamic_sym -> OctavPipeline on the caller's stream
        self.fq = {}         # parameter identity -> FakeQuantSet

    def accumulators(self, bins):
        a = self.acc.get(bins)
        if a is None:
            a = self.acc[bins] = ops.CalibAccumulators(self.plan.T, self.device, bins)
            a.ranges = torch.empty(a.n * ctypes.sizeof(_hip.HistRange), dtype=torch.uint8, device=self.device)
        return a

    def pipeline(self, dynamic_sym):
        p = self.pipes.get(bool(dynamic_sym))
        if p is None:
            p = self.pipes[bool(dynamic_sym)] = ops.OctavPipeline(bool(dynamic_sym), self.device, lanes=1)
        return p


def _cached(xs, batch=None):
    """The cached plan for this tensor set: xs[t] = tensor t of a batch of `batch` images (default: xs[0].shape[0])."""
    if not xs:
        raise ValueError("an empty tensor list")
    b = int(batch if batch is not None else (xs[0].shape[0] if xs[0].dim() > 0 else 1))
    if b < 1 or any(x.numel() % b for x in xs):     # (a per-tensor op passes batch = 1)
        raise ValueError(f"every tensor must hold a whole number of elements per image (batch {b})")
    dev = xs[0].device
    key = (tuple(x.numel() // b for x in xs), b, dev.index)
    c = _PLANS.get(key)
    if c is None:
        if len(_PLANS) >= _PLANS_MAX:
            _PLANS.pop(next(iter(_PLANS)))
        c = _PLANS[key] = _Cached(list(key[0]), b, dev)
    return c


def _contig(xs):
    return [x if x.is_contiguous() else x.contiguous() for x in xs]


# ---------------------------------------------------------------------------------------------- min / max
@torch.library.custom_op("dipoorlet::minmax", mutates_args=(), device_types="cuda")
def minmax(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    c = _cached([x], 1)
    a = c.accumulators(2048)
    a.reset_minmax()
    a.minmax_accumulate(c.plan, [x])
    lo, hi = a.finalize_minmax()
    return torch.cat([lo, hi])


@minmax.register_fake
def _(x):
    return x.new_empty(2, dtype=torch.float32)


@torch.library.custom_op("dipoorlet::minmax_batched", mutates_args=("mins", "maxs"), device_types="cuda")
def minmax_batched(xs: List[torch.Tensor], mins: torch.Tensor, maxs: torch.Tensor) -> None:
    xs = _contig(xs)
    c = _cached(xs, 1)          # (one slot per tensor: the images of a batch merge for free)
    if mins.numel() != c.plan.T or maxs.numel() != c.plan.T or mins.dtype != torch.float32 or maxs.dtype != torch.float32:
        raise ValueError("mins / maxs must be fp32 tensors with one entry per tensor of xs")
    a = c.accumulators(2048)
    a.reset_minmax()
    a.minmax_accumulate(c.plan, xs)
    lo, hi = a.finalize_minmax()
    # running form: NaN (from either side) propagates like numpy's min / max (torch.minimum / maximum propagate NaN)
    torch.minimum(mins, lo.view(mins.shape), out=mins)
    torch.maximum(maxs, hi.view(maxs.shape), out=maxs)


# ---------------------------------------------------------------------------------------------- histograms
def _hist_into(c, xs, bins, hist):
    a = c.accumulators(bins)
    L = _hip.lib()
    _hip.check(L.dpl_hist_prepare(ops._ptr(a.gmin), ops._ptr(a.gmax), a.n, a.bins, ops._ptr(a.ranges), ops._stream()), "dpl_hist_prepare")
    tab = c.plan.seg_table(xs)
    w = c.plan.work("hist")
    _hip.check(L.dpl_abs_hist_accumulate(*w.args(), ops._ptr(tab), ops._ptr(a.ranges), a.bins, ops._ptr(hist), ops._stream()),
               "dpl_abs_hist_accumulate")


@torch.library.custom_op("dipoorlet::abs_hist_", mutates_args=("hist",), device_types="cuda")
def abs_hist_(x: torch.Tensor, dmax: float, bins: int, hist: torch.Tensor) -> None:
    if hist.dtype != torch.int64 or hist.numel() != bins or not hist.is_contiguous():
        raise ValueError("hist must be a contiguous int64 tensor with `bins` entries")
    x = x.contiguous()
    c = _cached([x], 1)
    a = c.accumulators(int(bins))
    a.gmin.fill_(0.0)
    a.gmax.fill_(float(dmax))
    _hist_into(c, [x], int(bins), hist)


@torch.library.custom_op("dipoorlet::abs_hist_batched_", mutates_args=("hist",), device_types="cuda")
def abs_hist_batched_(xs: List[torch.Tensor], mins: torch.Tensor, maxs: torch.Tensor, bins: int, hist: torch.Tensor) -> None:
    xs = _contig(xs)
    c = _cached(xs, 1)
    T = c.plan.T
    if hist.dtype != torch.int64 or hist.numel() != T * bins or not hist.is_contiguous():
        raise ValueError("hist must be a contiguous int64 tensor [T, bins]")
    if mins.numel() != T or maxs.numel() != T:
        raise ValueError("mins / maxs must hold one entry per tensor of xs")
    a = c.accumulators(int(bins))
    a.gmin.copy_(mins.view(-1))
    a.gmax.copy_(maxs.view(-1))
    _hist_into(c, xs, int(bins), hist)


@torch.library.custom_op("dipoorlet::hist_percentile", mutates_args=(), device_types="cuda")
def hist_percentile(hist: torch.Tensor, gmin: float, gmax: float, threshold: float) -> torch.Tensor:
    if hist.dtype != torch.int64 or not hist.is_contiguous():
        raise ValueError("hist must be a contiguous int64 tensor")
    bins = hist.numel()
    rng = _RANGE.get(hist.device.index)
    if rng is None:
        rng = _RANGE[hist.device.index] = torch.empty(2, dtype=torch.float32, device=hist.device)
    rng[0].fill_(float(gmin))
    rng[1].fill_(float(gmax))
    clip = torch.empty(2, dtype=torch.float32, device=hist.device)
    _hip.check(_hip.lib().dpl_hist_percentile(ops._ptr(hist), ops._ptr(rng[0:1]), ops._ptr(rng[1:2]), 1, bins, float(threshold),
                                              ops._ptr(clip), ops._stream()), "dpl_hist_percentile")
    return clip


@hist_percentile.register_fake
def _(hist, gmin, gmax, threshold):
    return hist.new_empty(2, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- OCTAV
def _octav(xs, dynamic_sym, batch=None):
    xs = _contig(xs)
    c = _cached(xs, batch)
    pipe = c.pipeline(dynamic_sym)
    out = pipe.submit(c.plan, xs)
    ps = pipe._plans.get(id(c.plan))
    if ps is not None:      # (the exact-tail form: the rows are written on the pipeline's side stream — order this stream behind it)
        done = ps["sets"][(ps["calls"] - 1) % len(ps["sets"])]["done"]
        torch.cuda.current_stream(c.device).wait_event(done)
    return out


@torch.library.custom_op("dipoorlet::octav", mutates_args=(), device_types="cuda")
def octav(x: torch.Tensor, dynamic_sym: bool) -> torch.Tensor:
    return _octav([x], dynamic_sym, 1).reshape(3)


@octav.register_fake
def _(x, dynamic_sym):
    return x.new_empty(3, dtype=torch.float32)


@torch.library.custom_op("dipoorlet::octav_batched", mutates_args=(), device_types="cuda")
def octav_batched(xs: List[torch.Tensor], dynamic_sym: bool) -> torch.Tensor:
    return _octav(xs, dynamic_sym)


@octav_batched.register_fake
def _(xs, dynamic_sym):
    return xs[0].new_empty(xs[0].shape[0], len(xs), 3, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- weights, fake quant
@torch.library.custom_op("dipoorlet::rowwise_minmax", mutates_args=(), device_types="cuda")
def rowwise_minmax(w2d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.rowwise_minmax(w2d.contiguous())


@rowwise_minmax.register_fake
def _(w2d):
    return w2d.new_empty(w2d.shape[0]), w2d.new_empty(w2d.shape[0])


@torch.library.custom_op("dipoorlet::fake_quant", mutates_args=(), device_types="cuda")
def fake_quant(x: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, axis: int, qlo: int,
               qhi: int) -> torch.Tensor:
    return ops.fake_quant(x.contiguous(), scale, zero_point, qlo, qhi, axis=axis if scale.numel() > 1 else None)


@fake_quant.register_fake
def _(x, scale, zero_point, axis, qlo, qhi):
    return torch.empty_like(x)


@torch.library.custom_op("dipoorlet::fake_quant_relu", mutates_args=(), device_types="cuda")
def fake_quant_relu(x: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, axis: int, qlo: int,
                    qhi: int) -> torch.Tensor:
    return ops.fake_quant(x.contiguous(), scale, zero_point, qlo, qhi, axis=axis if scale.numel() > 1 else None, pre="relu")


@fake_quant_relu.register_fake
def _(x, scale, zero_point, axis, qlo, qhi):
    return torch.empty_like(x)


@torch.library.custom_op("dipoorlet::fake_quant_add_relu", mutates_args=(), device_types="cuda")
def fake_quant_add_relu(x: torch.Tensor, x2: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, axis: int, qlo: int,
                        qhi: int) -> torch.Tensor:
    return ops.fake_quant(x.contiguous(), scale, zero_point, qlo, qhi, axis=axis if scale.numel() > 1 else None, pre="add_relu",
                          x2=x2.contiguous())


@fake_quant_add_relu.register_fake
def _(x, x2, scale, zero_point, axis, qlo, qhi):
    return torch.empty_like(x)


@torch.library.custom_op("dipoorlet::fake_quant_set", mutates_args=(), device_types="cuda")
def fake_quant_set(xs: List[torch.Tensor], scales: List[torch.Tensor], zero_points: List[torch.Tensor], inner: List[int],
                   qlo: List[int], qhi: List[int]) -> List[torch.Tensor]:
    """Fused Q -> DQ of every tensor of a batch in ONE launch (dpl_fake_quant_items).  scales[t] / zero_points[t]: [1] or [C] (fp32 /
    int32 device tensors — kept by the cached plan: pass the same tensors on every call); inner[t]: elements behind the channel
    axis of tensors[t] as laid out in memory (ignored per tensor)."""
    xs = _contig(xs)
    c = _cached(xs, 1)
    key = (tuple(s.data_ptr() for s in scales), tuple(z.data_ptr() for z in zero_points), tuple(inner), tuple(qlo), tuple(qhi))
    fq = c.fq.get(key)
    if fq is None:
        if len(c.fq) >= 8:
            c.fq.pop(next(iter(c.fq)))
        fq = c.fq[key] = ops.FakeQuantSet(c.plan, list(zip(scales, zero_points, inner, qlo, qhi)))
    return fq(xs)


@fake_quant_set.register_fake
def _(xs, scales, zero_points, inner, qlo, qhi):
    return [torch.empty_like(x) for x in xs]
