"""Sparse + quantised layer — counterpart of dipoorlet/weight_transform/sparse_quant_layer.py.

The layer's own weight is learned (SGD) through  prune (magnitude mask, recomputed every forward)  ->  fake
quantisation with a straight-through round.  The masks are a k-th-value / top-2-of-4 selection on a weight-sized
tensor (torch on the device); the quantiser, its gradient and the SGD update are the fused kernels of
csrc/round_kernels.hip (dpl_sparse_quant, dpl_sparse_step).
"""
import math

import torch

from .. import _hip
from ..executor import _OPS
from ..ops import _ptr, _stream
from .ada_quant_layer import _OP_CTX, _channel_layout, _require_cuda

__all__ = ["SparseQLayer", "quant_weight_wo_roundmask", "prune_weight", "create_unstruction_mask", "create_nv24_mask"]


def create_unstruction_mask(weight, sparsity):
    """sparse_quant_layer.py:32-40 — drop the `sparsity` fraction of smallest magnitudes (everything tied with the
    threshold goes too)."""
    a = weight.detach().abs()
    prune_num = int(sparsity * a.numel())
    if prune_num == 0:
        return torch.ones_like(a)
    threshold = torch.kthvalue(a.reshape(-1), prune_num).values      # = topk(prune_num, largest=False).max()
    return (a > threshold).to(a.dtype)


def create_nv24_mask(weight, N=2, M=4):
    """sparse_quant_layer.py:43-56 — keep the N largest magnitudes of every M consecutive input channels."""
    a = weight.detach().abs()
    if a.dim() == 4:
        t = a.permute(0, 2, 3, 1).reshape(-1, M)
    elif a.dim() == 2:
        t = a.reshape(-1, M)
    else:
        raise ValueError("nv24 needs a 2-D or 4-D weight")
    drop = torch.argsort(t, dim=1)[:, :M - N]
    mask = torch.ones_like(t).scatter_(1, drop, 0.0)
    if a.dim() == 4:
        return mask.reshape(a.shape[0], a.shape[2], a.shape[3], a.shape[1]).permute(0, 3, 1, 2).contiguous()
    return mask.reshape(a.shape)


def prune_mask(weight, sparse_info):
    if sparse_info["pattern"] == "unstruction":
        return create_unstruction_mask(weight, sparse_info["rate"])
    if sparse_info["pattern"] == "nv24":
        return create_nv24_mask(weight, 2, 4)
    raise ValueError(f"unknown sparsity pattern {sparse_info['pattern']!r}")


def prune_weight(weight, sparse_info):
    """sparse_quant_layer.py:59-64."""
    return weight * prune_mask(weight, sparse_info)


def quant_weight_wo_roundmask(weight, scale, q_min, q_max, per_channel, mask=None):
    """sparse_quant_layer.py:21-29 (forward value): clamp?(round(w / scale)) * scale, channel axis 0; the clamp only
    on the per-channel branch, like the reference."""
    _require_cuda(weight, "weight")
    w = weight.detach().contiguous()
    scale, q_min, q_max = (t.reshape(-1).contiguous().float() for t in (scale, q_min, q_max))
    nch, inner = _channel_layout(w.numel(), scale)
    out = torch.empty_like(w)
    _hip.check(_hip.lib().dpl_sparse_quant(_ptr(w), _ptr(mask.contiguous()) if mask is not None else None, _ptr(scale),
                                           _ptr(q_min), _ptr(q_max), w.numel(), nch, inner, 1 if per_channel else 0,
                                           _ptr(out), _stream()), "dpl_sparse_quant")
    return out


class SparseQLayer:
    """sparse_quant_layer.py:69-175 — a Conv / Gemm / ConvTranspose node whose weight is learned under a magnitude
    mask and a fake quantiser.  `qw` is the leaf autograd deposits dL/d(qw) into; step() turns it into the SGD update
    of the weight (one kernel)."""

    def __init__(self, node, weight, bias, qw_tensor, relu_flag, sparse_info, lr=1e-3, momentum=0.9, weight_decay=1e-4):
        self.node, self.type = node, node.op_type
        self.transposed = self.type == "ConvTranspose"
        w = weight.transpose(0, 1) if self.transposed else weight
        self.weight = w.detach().contiguous().float().clone()          # channel first (sparse_quant.py:68-69)
        self.scale, self.q_min, self.q_max = (qw_tensor[k].reshape(-1).contiguous().float()
                                              for k in ("scale", "q_min", "q_max"))
        self.n = self.weight.numel()
        self.nch, self.inner = _channel_layout(self.n, self.scale)
        self.clamp = 1 if qw_tensor["per_channel"] else 0
        self.bias, self.relu_flag, self.sparse_info = bias, relu_flag, sparse_info
        self.base_lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.buf = torch.zeros_like(self.weight)
        self.steps = 0
        self.mask = None
        self.qw = torch.empty_like(self.weight).requires_grad_(True)

    def _quant(self, out, mask):
        _hip.check(_hip.lib().dpl_sparse_quant(_ptr(self.weight), _ptr(mask), _ptr(self.scale), _ptr(self.q_min),
                                               _ptr(self.q_max), self.n, self.nch, self.inner, self.clamp, _ptr(out),
                                               _stream()), "dpl_sparse_quant")

    def __call__(self, x, apply_relu=True):
        with torch.no_grad():
            self.mask = prune_mask(self.weight, self.sparse_info)
            self._quant(self.qw, self.mask)
        qw = self.qw.transpose(0, 1) if self.transposed else self.qw
        args = (x, qw) if self.bias is None else (x, qw, self.bias)
        x = _OPS[self.type](_OP_CTX, self.node, *args)
        return torch.relu(x) if (self.relu_flag and apply_relu) else x

    def step(self, lr, grad_scale=1.0, grad_out=None):
        """Consume qw.grad: straight-through gradient under the mask + SGD(momentum, weight decay) — one kernel."""
        g = self.qw.grad.contiguous()
        _hip.check(_hip.lib().dpl_sparse_step(_ptr(g), _ptr(self.weight), _ptr(self.mask), _ptr(self.buf),
                                              _ptr(self.scale), _ptr(self.q_min), _ptr(self.q_max), self.n, self.nch,
                                              self.inner, self.clamp, grad_scale, lr, self.momentum, self.weight_decay,
                                              1 if self.steps == 0 else 0, 1,
                                              _ptr(grad_out) if grad_out is not None else None, _stream()),
                   "dpl_sparse_step")
        self.steps += 1
        self.qw.grad = None

    def new_weight(self):
        """sparse_quant.py:93-101 — quantise(prune(learned weight)), in the graph's own layout."""
        out = torch.empty_like(self.weight)
        self._quant(out, prune_mask(self.weight, self.sparse_info))
        return out.transpose(0, 1).contiguous() if self.transposed else out


def cosine_lr(base_lr, epoch, t_max, eta_min=0.0):
    """torch.optim.lr_scheduler.CosineAnnealingLR (closed form) for the learning rate of `epoch`."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
