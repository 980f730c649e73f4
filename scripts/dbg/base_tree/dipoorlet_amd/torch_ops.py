"""`torch.ops.dipoorlet.*` — the custom-op spelling of the kernel library, for callers that live in
torch (ORT-IOBinding / eager pipelines, torch.compile graphs).  Thin registrations over dipoorlet_amd.ops: every op
runs the HIP kernels on the current stream and is registered for the 'cuda' (ROCm) device only — there is no CPU
implementation to fall back to.

    dipoorlet::minmax(Tensor x) -> Tensor                      [2] fp32 (min, max); NaN if x holds one
    dipoorlet::minmax_batched(Tensor[] xs, Tensor(a!) mins, Tensor(b!) maxs) -> ()   running min / max per tensor
    dipoorlet::abs_hist_(Tensor x, float dmax, int bins, Tensor(a!) hist) -> ()      hist += np.histogram(|x|, bins, (0, dmax))
    dipoorlet::hist_percentile(Tensor hist, float gmin, float gmax, float threshold) -> Tensor   [2] fp32 clip
    dipoorlet::octav(Tensor x, bool dynamic_sym) -> Tensor     [3] fp32 (s, min, max) (forward_net.py:315-330)
    dipoorlet::rowwise_minmax(Tensor w2d) -> (Tensor, Tensor)
    dipoorlet::fake_quant(Tensor x, Tensor scale, Tensor zero_point, int axis, int qlo, int qhi) -> Tensor
"""
from typing import List, Tuple

import torch

from . import ops


@torch.library.custom_op("dipoorlet::minmax", mutates_args=(), device_types="cuda")
def minmax(x: torch.Tensor) -> torch.Tensor:
    return ops.minmax(x.contiguous())


@minmax.register_fake
def _(x):
    return x.new_empty(2, dtype=torch.float32)


@torch.library.custom_op("dipoorlet::minmax_batched", mutates_args=("mins", "maxs"), device_types="cuda")
def minmax_batched(xs: List[torch.Tensor], mins: torch.Tensor, maxs: torch.Tensor) -> None:
    xs = [x.contiguous() for x in xs]
    plan = ops.TensorSetPlan([x.numel() for x in xs], 1, xs[0].device)
    acc = ops.CalibAccumulators(len(xs), xs[0].device)
    acc.minmax_accumulate(plan, xs)
    lo, hi = acc.finalize_minmax()
    # running form: NaN (from either side) propagates like numpy's min / max
    mins.copy_(torch.minimum(mins, lo))   # torch.minimum / maximum propagate NaN
    maxs.copy_(torch.maximum(maxs, hi))


@torch.library.custom_op("dipoorlet::abs_hist_", mutates_args=("hist",), device_types="cuda")
def abs_hist_(x: torch.Tensor, dmax: float, bins: int, hist: torch.Tensor) -> None:
    if hist.dtype != torch.int64 or hist.numel() != bins:
        raise ValueError("hist must be an int64 tensor with `bins` entries")
    h, _ = ops.abs_hist(x.contiguous(), bins, 0.0, dmax)
    hist.add_(h.reshape(hist.shape))


@torch.library.custom_op("dipoorlet::hist_percentile", mutates_args=(), device_types="cuda")
def hist_percentile(hist: torch.Tensor, gmin: float, gmax: float, threshold: float) -> torch.Tensor:
    acc = ops.CalibAccumulators(1, hist.device, hist.numel())
    acc.set_minmax(torch.tensor([gmin], dtype=torch.float32, device=hist.device),
                   torch.tensor([gmax], dtype=torch.float32, device=hist.device))
    acc.hist_prepare()
    acc.hist.copy_(hist.reshape(1, -1))
    return acc.hist_percentile(threshold)[0]


@hist_percentile.register_fake
def _(hist, gmin, gmax, threshold):
    return hist.new_empty(2, dtype=torch.float32)


@torch.library.custom_op("dipoorlet::octav", mutates_args=(), device_types="cuda")
def octav(x: torch.Tensor, dynamic_sym: bool) -> torch.Tensor:
    x = x.contiguous()
    plan = ops.TensorSetPlan([x.numel()], 1, x.device)
    return ops.octav_batch(plan, [x], dynamic_sym)[0, 0]


@octav.register_fake
def _(x, dynamic_sym):
    return x.new_empty(3, dtype=torch.float32)


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
