"""CPU ORACLE (test infrastructure, NOT product code) — restatement of the reference's AdaRound / BRECQ / QDrop
arithmetic with every gradient written out by hand.

  * Only tests/ may import this module; the product package never does.
  * The reference expresses these as torch expressions and lets autograd differentiate them
    (dipoorlet/weight_transform/ada_quant_layer.py:28-125, adaround.py:119-144, torch.optim.Adam).  Here the
    forward values AND the derivatives autograd would produce are explicit numpy fp32 formulas — including the
    corners that matter for parity: clamp() passes its gradient on the closed interval, maximum / minimum split
    it on ties, pow(x, 0.0) has zero gradient, round() has none — so that the HIP kernels can be checked against
    something that is not the same autograd graph.
  * Parity pin: tests/test_round_oracle_golden.py checks every function against tests/golden/round_level.*,
    produced by tests/golden/gen_golden_round.py from the reference's own code on CPU torch.
  * Matrix products / convolutions inside `train_layer` use torch's CPU kernels (third-party numerics; the
    reference uses the same library on the GPU).
"""
import math

import numpy as np

F32 = np.float32
ZETA, GAMMA = 1.1, -0.1
ZG = F32(ZETA - GAMMA)      # torch casts the python scalar to the tensor dtype
G32 = F32(GAMMA)


def sigmoid(a):
    a = np.asarray(a, F32)
    return (F32(1) / (F32(1) + np.exp(-a, dtype=F32))).astype(F32)


def rect_sigmoid(mask):
    """ada_quant_layer.py:105-106 -> (h, dh/dmask).  clamp(0, 1) passes the gradient where 0 <= raw <= 1."""
    sg = sigmoid(mask)
    raw = (ZG * sg + G32).astype(F32)
    h = np.clip(raw, F32(0), F32(1)).astype(F32)
    inside = (raw >= 0) & (raw <= 1)
    dh = np.where(inside, (ZG * (F32(1) - sg)).astype(F32) * sg, F32(0)).astype(F32)
    return h, dh


def _bc(v, ndim):
    v = np.asarray(v, F32).reshape(-1)
    return v.reshape([-1] + [1] * (ndim - 1)) if v.size > 1 else v.reshape([1] * ndim)


def alpha_init(w, scale):
    """adaround.py:67 + ada_quant_layer.py:147: rest = w/s - floor(w/s); mask = -log((zeta-gamma)/(rest-gamma) - 1).
    torch evaluates `scalar / tensor` as reciprocal(tensor) * scalar."""
    w = np.asarray(w, F32)
    t = (w / _bc(scale, w.ndim)).astype(F32)
    wf = np.floor(t).astype(F32)
    rest = (t - wf).astype(F32)
    inv = (F32(1) / (rest - G32).astype(F32)).astype(F32)
    return wf, (-np.log((inv * ZG).astype(F32) - F32(1), dtype=F32)).astype(F32)


def _clamp_with_pass(v, q_min, q_max):
    """torch.max(v, q_min) then torch.min(., q_max) and the factor autograd applies to the incoming gradient."""
    f_lo = np.where(v > q_min, F32(1), np.where(v == q_min, F32(0.5), F32(0)))
    v1 = np.maximum(v, q_min)
    f_hi = np.where(v1 < q_max, F32(1), np.where(v1 == q_max, F32(0.5), F32(0)))
    return np.minimum(v1, q_max).astype(F32), (f_lo * f_hi).astype(F32)


def quant_weight(w, mask, scale, q_min, q_max, per_channel, soft=True):
    """ada_quant_layer.py:39-50 -> (quantised weight, d(quantised weight)/d(mask)).  Only the per-channel branch
    clamps (the per-tensor branch calls weight.clamp(...) without using the result)."""
    w = np.asarray(w, F32)
    s = _bc(scale, w.ndim)
    wf = np.floor((w / s).astype(F32)).astype(F32)
    if soft:
        h, dh = rect_sigmoid(mask)
    else:
        h, dh = (np.asarray(mask, F32) >= 0).astype(F32), np.zeros_like(w)
    v = (wf + h).astype(F32)
    passf = np.ones_like(w)
    if per_channel:
        v, passf = _clamp_with_pass(v, _bc(q_min, w.ndim), _bc(q_max, w.ndim))
    return (v * s).astype(F32), (np.broadcast_to(s, w.shape) * passf * dh).astype(F32)


def temp_decay(t, t_max, rel_start_decay=0.2, start_b=20, end_b=2):
    """ada_quant_layer.py:119-134."""
    start = rel_start_decay * t_max
    if t < start:
        return 0.0
    rel_t = (t - start) / (t_max - start)
    return end_b + 0.5 * (start_b - end_b) * (1 + np.cos(rel_t * np.pi))


def reg_value_grad(mask, beta, lam=0.01):
    """ada_quant_layer.py:108-110: lam * sum(1 - (|h - 0.5| * 2)^beta) and its gradient w.r.t. the mask.
    beta == 0: pow(x, 0) = 1 everywhere and torch defines its gradient as zero."""
    h, dh = rect_sigmoid(mask)
    if beta == 0.0:
        return 0.0, np.zeros_like(h)
    b = F32(beta)
    d = (h - F32(0.5)).astype(F32)
    u = (np.abs(d) * F32(2)).astype(F32)
    val = lam * float(np.sum((F32(1) - np.power(u, b, dtype=F32)).astype(np.float64)))
    with np.errstate(divide="ignore", invalid="ignore"):
        dp = np.where(u > 0, b * np.power(u, b - F32(1), dtype=F32), F32(0)).astype(F32)
    g = (F32(-lam) * dp * F32(2) * np.sign(d).astype(F32) * dh).astype(F32)
    return val, g


def l2_value_grad(pred, tgt, relu=False):
    """ada_quant_layer.py:115-116: ((pred - tgt)^2).sum(1).mean() and d/d(pred); with relu the loss is taken on
    max(pred, 0) and the ReLU's gradient (pred > 0) is folded in."""
    pred, tgt = np.asarray(pred, F32), np.asarray(tgt, F32)
    m = pred.size // pred.shape[1]
    y = np.maximum(pred, F32(0)) if relu else pred
    d = (y - tgt).astype(F32)
    val = float(np.sum((d * d).astype(np.float64))) / m
    g = (F32(F32(1.0 / m) * F32(2)) * d).astype(F32)
    if relu:
        g = np.where(pred > 0, g, F32(0)).astype(F32)
    return val, g


def quant_acti_drop(x, r, scale, q_min, q_max, prob):
    """ada_quant_layer.py:28-36 with the uniform draw `r` given -> (y, dy/dx as a 0/1 mask): round() blocks the
    gradient, the kept (un-quantised) elements pass it."""
    x = np.asarray(x, F32)
    q = np.rint((x / F32(scale)).astype(F32)).astype(F32)      # torch.round: half to even
    q = (np.minimum(np.maximum(q, F32(q_min)), F32(q_max)) * F32(scale)).astype(F32)
    if prob >= 1.0:
        return q, np.zeros_like(x)
    keep_q = np.asarray(r, F32) < F32(prob)
    return np.where(keep_q, q, x).astype(F32), np.where(keep_q, F32(0), F32(1)).astype(F32)


class Adam:
    """torch.optim.Adam, single-tensor form (lr 1e-3, betas (0.9, 0.999), eps 1e-8): lerp, addcmul, addcdiv with the
    python-double scalars cast to fp32 at the op."""

    def __init__(self, shape, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.m, self.v, self.t = np.zeros(shape, F32), np.zeros(shape, F32), 0
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps

    def step(self, p, g):
        self.t += 1
        g = np.asarray(g, F32)
        self.m = (self.m + F32(1 - self.b1) * (g - self.m).astype(F32)).astype(F32)
        self.v = ((self.v * F32(self.b2)).astype(F32) + (F32(1 - self.b2) * g).astype(F32) * g).astype(F32)
        step_size = self.lr / (1 - self.b1 ** self.t)
        bc2_sqrt = math.sqrt(1 - self.b2 ** self.t)
        denom = ((np.sqrt(self.v) / F32(bc2_sqrt)).astype(F32) + F32(self.eps)).astype(F32)
        return (p + ((F32(-step_size) * self.m).astype(F32) / denom).astype(F32)).astype(F32)


def train_layer(kind, w, b, x, fp, scale, q_min, q_max, per_channel, relu, bs, epochs, snapshots=()):
    """learning_round_mask (adaround.py:119-144) for ONE Gemm ('gemm': y = x W^T + b) or 3x3 / pad-1 Conv ('conv')
    layer, no DDP: returns (final mask, {step: mask}, hard-rounded weight)."""
    import torch
    import torch.nn.functional as F
    n = x.shape[0]
    nb = int(math.ceil(n / bs))
    total = epochs * nb
    _, mask = alpha_init(w, scale)
    opt = Adam(mask.shape)
    xt, bt = torch.from_numpy(np.asarray(x, F32)), torch.from_numpy(np.asarray(b, F32))
    snaps, cur = {}, 0
    for _ in range(epochs):
        for i in range(nb):
            qw, dqw = quant_weight(w, mask, scale, q_min, q_max, per_channel)
            qwt = torch.from_numpy(qw).requires_grad_(True)
            xb = xt[i * bs:(i + 1) * bs]
            z = F.linear(xb, qwt, bt) if kind == "gemm" else F.conv2d(xb, qwt, bt, 1, 1)
            _, gz = l2_value_grad(z.detach().numpy(), fp[i * bs:(i + 1) * bs], relu)
            z.backward(torch.from_numpy(gz))
            _, greg = reg_value_grad(mask, temp_decay(cur, total))
            g = (qwt.grad.numpy() * dqw).astype(F32) + greg
            mask = opt.step(mask, g)
            cur += 1
            if cur in snapshots:
                snaps[cur] = mask.copy()
    hard, _ = quant_weight(w, mask, scale, q_min, q_max, per_channel, soft=False)
    return mask, snaps, hard
