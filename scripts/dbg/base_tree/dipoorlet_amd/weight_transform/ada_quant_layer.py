"""The learnable-rounding layer on MI355X — counterpart of dipoorlet/weight_transform/ada_quant_layer.py.

The reference expresses the soft quantiser, the rounding regulariser, the L2 reconstruction loss, QDrop and the
Adam update as eager torch ops (~60 kernel launches and a dozen weight- and activation-sized temporaries per
iteration).  Here a learning iteration is
    conv / gemm forward (MIOpen / hipBLASLt through torch)  ->  L2_norm (ONE kernel: loss + dL/dz, ReLU folded in)
    ->  conv backward (torch autograd)  ->  RoundingParam.step (ONE kernel: chain rule through the soft quantiser,
    regulariser value + gradient, Adam, and the next iteration's soft-quantised weight)
with the arithmetic of the reference's functions (goldens: tests/golden/round_level.*).  The public names follow
the reference so its callers read the same.
"""
import ctypes as C
import math
import types

import numpy as np
import torch

from .. import _hip
from ..executor import _OPS
from ..ops import _ptr, _stream

__all__ = ["quant_acti", "quant_weight", "adaround_reg", "TempDecay", "L2_norm", "RoundingParam", "RoundSchedule",
           "AdaQLayer"]


def _require_cuda(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise _hip.DipoorletHipError(f"{name} must be a float32 ROCm device tensor; dipoorlet_amd has no CPU path")


class TempDecay:
    """ada_quant_layer.py:119-134 — regulariser temperature: 0 for the first 20 % of the iterations, then a cosine
    decay from 20 to 2."""

    def __init__(self, t_max, rel_start_decay=0.2, start_b=20, end_b=2):
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b, self.end_b = start_b, end_b

    def __call__(self, t):
        if t < self.start_decay:
            return 0.0
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + 0.5 * (self.start_b - self.end_b) * (1 + np.cos(rel_t * np.pi))


def _channel_layout(n, scale):
    nch = int(scale.numel())
    if n % nch:
        raise _hip.DipoorletHipError(f"{n} weight elements do not split into {nch} channels")
    return nch, n // nch


def _step_params(step=1, adam=0, clamp=0, grad_scale=1.0, reg_beta=0.0, reg_lambda=0.01, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8):
    return _hip.RoundStepParams(lr, betas[0], betas[1], eps, step, adam, clamp, 0, grad_scale, reg_beta, reg_lambda, 0.0)


class adaround_reg:
    """ada_quant_layer.py:96-112 — lambda * sum(1 - |2 h(mask) - 1|^beta), beta annealed by TempDecay."""

    def __init__(self, max_iter=10000, zeta=1.1, gamma=-0.1, alpha=0.01, beta=20):
        if (zeta, gamma) != (1.1, -0.1):
            raise NotImplementedError("the kernels are built for the reference's zeta = 1.1, gamma = -0.1")
        self.zeta, self.gamma, self.alpha, self.beta = zeta, gamma, alpha, beta
        self.temp_anneal = TempDecay(max_iter)

    def rectified_sigmoid(self, round_mask):
        return ((self.zeta - self.gamma) * torch.sigmoid(round_mask) + self.gamma).clamp(0, 1)

    def value_and_grad(self, round_mask, iter):
        """(value as an fp64 device scalar, d value / d round_mask) in one kernel."""
        _require_cuda(round_mask, "round_mask")
        self.beta = self.temp_anneal(iter)
        mask = round_mask.detach().contiguous()
        grad = torch.empty_like(mask)
        val = torch.zeros(1, dtype=torch.float64, device=mask.device)
        one = torch.ones(1, dtype=torch.float32, device=mask.device)
        p = _step_params(reg_beta=float(self.beta), reg_lambda=self.alpha)
        _hip.check(_hip.lib().dpl_round_step(None, _ptr(mask), _ptr(mask), None, None, _ptr(one), None, None,
                                             mask.numel(), 1, mask.numel(), C.byref(p), None, None, _ptr(grad),
                                             _ptr(val), _stream()), "dpl_round_step")
        return val[0], grad

    def __call__(self, round_mask, iter):
        return self.value_and_grad(round_mask, iter)[0]

    forward = __call__


def L2_norm(pred, tgt, relu=False, loss=None):
    """ada_quant_layer.py:115-116 — ((pred - tgt)^2).sum(1).mean(), fused with its gradient:
    returns (loss as an fp64 device tensor [1], d loss / d pred).  relu=True evaluates the loss on max(pred, 0)
    and folds the ReLU's gradient in (pred is then the pre-activation).  `loss`, if given, is accumulated into."""
    _require_cuda(pred, "pred")
    pred, tgt = pred.detach().contiguous(), tgt.contiguous()
    if pred.shape != tgt.shape:
        raise _hip.DipoorletHipError(f"L2_norm: shapes differ {tuple(pred.shape)} vs {tuple(tgt.shape)}")
    m = pred.numel() // pred.shape[1]          # .sum(1) then .mean() over the rest
    grad = torch.empty_like(pred)
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float64, device=pred.device)
    _hip.check(_hip.lib().dpl_l2_loss(_ptr(pred), _ptr(tgt), pred.numel(), 1 if relu else 0,
                                      float(np.float32(1.0 / m) * np.float32(2.0)), 1.0 / m, _ptr(grad), _ptr(loss),
                                      _stream()), "dpl_l2_loss")
    return loss, grad


class _ActiDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, q_min, q_max, prob):
        x = x.contiguous()
        r = torch.rand_like(x) if prob < 1.0 else None
        y = torch.empty_like(x)
        _hip.check(_hip.lib().dpl_acti_drop_fwd(_ptr(x), _ptr(r) if r is not None else None, x.numel(), scale, q_min,
                                                q_max, prob, _ptr(y), _stream()), "dpl_acti_drop_fwd")
        ctx.r, ctx.prob = r, prob
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        _hip.check(_hip.lib().dpl_acti_drop_bwd(_ptr(ctx.r) if ctx.r is not None else None, _ptr(gy), gy.numel(),
                                                ctx.prob, _ptr(gx), _stream()), "dpl_acti_drop_bwd")
        return gx, None, None, None, None


def _scalar(v):
    return float(v.reshape(-1)[0]) if isinstance(v, (torch.Tensor, np.ndarray)) else float(v)


def quant_acti(x, scale, q_min, q_max, prob, rand=None):
    """ada_quant_layer.py:28-36 — per-tensor fake quantisation with QDrop mixing; differentiable the way the
    reference's expression is under autograd (zero through round(), identity where the value was kept).
    `rand` (tests) prescribes the uniform draw."""
    _require_cuda(x, "x")
    if rand is not None:
        y = torch.empty_like(x)
        _hip.check(_hip.lib().dpl_acti_drop_fwd(_ptr(x.contiguous()), _ptr(rand.contiguous()), x.numel(),
                                                _scalar(scale), _scalar(q_min), _scalar(q_max), float(prob), _ptr(y),
                                                _stream()), "dpl_acti_drop_fwd")
        return y
    return _ActiDrop.apply(x, _scalar(scale), _scalar(q_min), _scalar(q_max), float(prob))


def quant_weight(weight, round_mask, scale, q_min, q_max, per_channel, soft=True):
    """ada_quant_layer.py:39-50 — (floor(w / scale) + h(mask)) * scale, clamped to [q_min, q_max] only on the
    per-channel branch (the reference's per-tensor branch discards its clamp).  Channel axis 0."""
    _require_cuda(weight, "weight")
    w = weight.detach().contiguous()
    scale, q_min, q_max = (t.reshape(-1).contiguous().float() for t in (scale, q_min, q_max))
    nch, inner = _channel_layout(w.numel(), scale)
    wfloor, alpha0 = torch.empty_like(w), torch.empty_like(w)
    L = _hip.lib()
    _hip.check(L.dpl_round_init(_ptr(w), _ptr(scale), w.numel(), nch, inner, _ptr(wfloor), _ptr(alpha0), _stream()),
               "dpl_round_init")
    out = torch.empty_like(w)
    _hip.check(L.dpl_round_quant(_ptr(wfloor), _ptr(round_mask.detach().contiguous()), _ptr(scale), _ptr(q_min),
                                 _ptr(q_max), w.numel(), nch, inner, 1 if per_channel else 0, 1 if soft else 0,
                                 _ptr(out), _stream()), "dpl_round_quant")
    return out


class RoundSchedule:
    """The learner's iteration counter, regulariser temperature and Adam corrections in device memory
    (dpl_round_sched), advanced by a one-thread kernel — what makes an iteration replayable as a hipGraph."""

    def __init__(self, t_max, device, lr=1e-3, betas=(0.9, 0.999)):
        self.t_max, self.lr, self.betas = int(t_max), lr, betas
        self.buf = torch.zeros(6, dtype=torch.int32, device=device)     # 24 bytes, zero = nothing done yet

    def advance(self):
        _hip.check(_hip.lib().dpl_round_sched_advance(_ptr(self.buf), self.t_max, self.lr, self.betas[0], self.betas[1],
                                                      _stream()), "dpl_round_sched_advance")

    def state(self):
        """HOST (synchronises): (iterations done, Adam steps done, temperature of the last iteration)."""
        raw = self.buf.cpu()
        return int(raw[0]), int(raw[1]), float(raw[2:3].view(torch.float32)[0])


class RoundingParam:
    """Rounding state of one layer, resident in HBM: floor(w / scale), the round mask, Adam moments and the
    current soft-quantised weight (a leaf that autograd deposits dL/d(qw) into).  Weight layout: channel first."""

    def __init__(self, weight, scale, q_min, q_max, per_channel, lr=1e-3):
        _require_cuda(weight, "weight")
        w = weight.detach().contiguous().float()
        self.shape = tuple(w.shape)
        self.scale, self.q_min, self.q_max = (t.reshape(-1).contiguous().float() for t in (scale, q_min, q_max))
        self.n = w.numel()
        self.nch, self.inner = _channel_layout(self.n, self.scale)
        self.clamp = 1 if per_channel else 0
        self.per_channel = bool(per_channel)
        self.lr = lr
        self.wfloor = torch.empty_like(w)
        self.round_mask = torch.empty_like(w)
        self.exp_avg = torch.zeros_like(w)
        self.exp_avg_sq = torch.zeros_like(w)
        self.steps = 0
        L = _hip.lib()
        _hip.check(L.dpl_round_init(_ptr(w), _ptr(self.scale), self.n, self.nch, self.inner, _ptr(self.wfloor),
                                    _ptr(self.round_mask), _stream()), "dpl_round_init")
        self.qw = torch.empty_like(w)
        self._quant(self.qw, soft=1)
        self.qw.requires_grad_(True)

    def _quant(self, out, soft):
        _hip.check(_hip.lib().dpl_round_quant(_ptr(self.wfloor), _ptr(self.round_mask), _ptr(self.scale),
                                              _ptr(self.q_min), _ptr(self.q_max), self.n, self.nch, self.inner,
                                              self.clamp, soft, _ptr(out), _stream()), "dpl_round_quant")

    def hard_weight(self):
        """quant_weight(..., soft=False): floor + (mask >= 0)."""
        out = torch.empty(self.shape, dtype=torch.float32, device=self.wfloor.device)
        self._quant(out, soft=0)
        return out

    def step(self, reg_beta, reg_lambda=0.01, grad_scale=1.0, reg_loss=None, grad_out=None, sched=None):
        """Consume qw.grad: mask gradient (+ regulariser), Adam update, refreshed soft weight — one kernel.
        sched: device schedule (RoundSchedule.buf) supplying the temperature and Adam corrections instead of the
        host (hipGraph replay)."""
        g = self.qw.grad
        self.steps += 1
        p = _step_params(step=self.steps, adam=1, clamp=self.clamp, grad_scale=grad_scale, reg_beta=float(reg_beta),
                         reg_lambda=reg_lambda, lr=self.lr)
        _hip.check(_hip.lib().dpl_round_step(_ptr(g.contiguous()) if g is not None else None, _ptr(self.wfloor),
                                             _ptr(self.round_mask), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                             _ptr(self.scale), _ptr(self.q_min), _ptr(self.q_max), self.n, self.nch,
                                             self.inner, C.byref(p), _ptr(sched) if sched is not None else None,
                                             _ptr(self.qw),
                                             _ptr(grad_out) if grad_out is not None else None,
                                             _ptr(reg_loss) if reg_loss is not None else None, _stream()),
                   "dpl_round_step")
        self.qw.grad = None

    def rounding_summary(self):
        """(ceil, floor, total) counts of the rectified sigmoid, as logged by adaround.py:136-143."""
        h = adaround_reg().rectified_sigmoid(self.round_mask)
        return int((h + 1e-4 >= 1.0).sum()), int((h <= 1e-4).sum()), h.numel()


_OP_CTX = types.SimpleNamespace(batch=1)


class AdaQLayer:
    """ada_quant_layer.py:137-252 — one Conv / Gemm / ConvTranspose node with a learnable rounding of its weight,
    an optional ReLU and an optional QDrop fake quantisation of its output.  The node's own attributes drive the
    same executor op the calibration forward uses, so layer and graph agree by construction."""

    def __init__(self, node, weight, bias, qw_tensor, qi_tensor, relu_flag, acti_quant, drop_ratio=0.5):
        if qw_tensor.get("type", "Linear") != "Linear":
            raise NotImplementedError("only 'Linear' weight quantisation is built (the nnie log-domain grid is not)")
        self.node, self.type = node, node.op_type
        self.transposed = self.type == "ConvTranspose"
        w = weight.transpose(0, 1).contiguous() if self.transposed else weight   # channel first (adaround.py:58-59)
        self.rp = RoundingParam(w, qw_tensor["scale"], qw_tensor["q_min"], qw_tensor["q_max"], qw_tensor["per_channel"])
        self.bias = bias
        self.relu_flag = relu_flag
        self.qi_tensor = qi_tensor
        self.acti_quant = bool(acti_quant) and qi_tensor is not None
        # per-tensor activation grid as host scalars, read back ONCE (never inside the learning loop)
        self._qi = tuple(_scalar(qi_tensor[k]) for k in ("scale", "q_min", "q_max")) if self.acti_quant else None
        self.drop_ratio = drop_ratio

    @property
    def round_mask(self):
        return self.rp.round_mask

    def __call__(self, x, apply_relu=True):
        qw = self.rp.qw.transpose(0, 1) if self.transposed else self.rp.qw
        args = (x, qw) if self.bias is None else (x, qw, self.bias)
        x = _OPS[self.type](_OP_CTX, self.node, *args)
        if self.relu_flag and apply_relu:
            x = torch.relu(x)
        if self.acti_quant:
            x = _ActiDrop.apply(x, self._qi[0], self._qi[1], self._qi[2], float(self.drop_ratio))
        return x

    def new_weight(self):
        """The hard-rounded weight in the graph's own layout."""
        w = self.rp.hard_weight()
        return w.transpose(0, 1).contiguous() if self.transposed else w


def total_iterations(epochs, n_images, batch_size, n_layers=1):
    """adaround.py:87 / brecq.py:62."""
    return epochs * n_layers * math.ceil(n_images / batch_size)
