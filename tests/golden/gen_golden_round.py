#!/usr/bin/env python3
"""Golden vectors for the AdaRound / BRECQ / QDrop inner loop, produced by running the
REFERENCE's own functions on CPU torch (imported from /root/reference under the same third-party stubs as
gen_golden.py; build container only):

  quant_weight, quant_acti, adaround_reg(.rectified_sigmoid/.forward), TempDecay, L2_norm
                                                       dipoorlet/weight_transform/ada_quant_layer.py:28-125
  round-mask initialisation                            ada_quant_layer.py:147 (with adaround.py:67)
  the training loop of learning_round_mask             adaround.py:119-144 (Adam, L2 + regulariser), run here
                                                       without DDP on a small Gemm and a small Conv layer

Only seeds/inputs and the reference's OUTPUTS (values, autograd gradients, trajectories) are stored.
Run:  python tests/golden/gen_golden_round.py      (needs /root/reference; not run on the GPU box)
"""
import json
import os
import sys
from unittest import mock

import numpy as np
import torch
import torch._dynamo  # noqa: F401  (before the stubs: it probes for 'onnx' with importlib at import time)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    for m in ["onnx", "onnx.helper", "onnx.numpy_helper", "onnx.external_data_helper", "onnxruntime",
              "onnxruntime.quantization", "onnxruntime.quantization.onnx_quantizer",
              "onnxruntime.quantization.quant_utils", "onnxsim", "termcolor"]:
        sys.modules[m] = mock.MagicMock(name=m)
    sys.path.insert(0, REF)
    import dipoorlet.weight_transform.ada_quant_layer as aq
    return aq


def weights(seed, shape, spread=0.05):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * spread


def channel_view(v, ndim):
    return v.reshape([-1] + [1] * (ndim - 1))


def quant_weight_cases(aq, out, meta):
    """Forward values (soft / hard) and d(sum(qw * G))/d(round_mask) through the reference's quant_weight."""
    for ci, (shape, per_channel, tight) in enumerate([((8, 4, 3, 3), True, False), ((8, 4, 3, 3), True, True),
                                                      ((6, 16), True, False), ((6, 16), False, False),
                                                      ((5, 3, 1, 1), False, True)]):
        w = weights(100 + ci, shape)
        g = torch.Generator().manual_seed(200 + ci)
        mask = torch.randn(shape, generator=g) * 3.0
        mask.view(-1)[::7] = 9.0     # sigmoid saturates: rectified sigmoid clamps at 1
        mask.view(-1)[3::11] = -9.0  # ... and at 0
        G = torch.randn(shape, generator=g)
        if per_channel:
            amax = w.abs().reshape(shape[0], -1).max(1).values
            scale = channel_view(amax / (200.0 if tight else 127.0), len(shape))   # tight: the clamp is active
            q_min = channel_view(torch.full((shape[0],), -127.0), len(shape))
            q_max = channel_view(torch.full((shape[0],), 127.0), len(shape))
        else:
            scale = (w.abs().max() / (200.0 if tight else 127.0)).reshape(())
            q_min, q_max = torch.tensor(-127.0), torch.tensor(127.0)
        key = f"qw{ci}"
        m = mask.clone().requires_grad_(True)
        soft = aq.quant_weight(w, m, scale, q_min, q_max, per_channel, soft=True)
        (soft * G).sum().backward()
        hard = aq.quant_weight(w, mask, scale, q_min, q_max, per_channel, soft=False)
        rest = (w / scale) - (w / scale).floor()
        reg = aq.adaround_reg()
        alpha0 = -torch.log((reg.zeta - reg.gamma) / (rest - reg.gamma) - 1)   # ada_quant_layer.py:147
        out[key + "_w"], out[key + "_mask"], out[key + "_G"] = w.numpy(), mask.numpy(), G.numpy()
        out[key + "_scale"] = scale.reshape(-1).numpy()
        out[key + "_soft"], out[key + "_hard"] = soft.detach().numpy(), hard.numpy()
        out[key + "_grad"] = m.grad.numpy()
        out[key + "_alpha0"] = alpha0.numpy()
        meta.append({"key": key, "shape": list(shape), "per_channel": per_channel, "tight": tight})


def reg_cases(aq, out, meta):
    """adaround_reg(max_iter).forward(mask, it): value, temperature and gradient over the decay schedule."""
    g = torch.Generator().manual_seed(7)
    mask = torch.randn(4096, generator=g) * 2.5
    mask[::13] = 9.0
    mask[5::17] = -9.0
    mask[1] = 0.0   # h = 0.5: |2h - 1| = 0
    out["reg_mask"] = mask.numpy()
    rows = []
    for max_iter, it in [(1000, 0), (1000, 199), (1000, 200), (1000, 201), (1000, 600), (1000, 999), (1000, 1000),
                         (60, 11), (60, 12), (60, 40)]:
        reg = aq.adaround_reg(max_iter)
        m = mask.clone().requires_grad_(True)
        v = reg(m, it)
        v.backward()
        out[f"reg_grad_{max_iter}_{it}"] = m.grad.numpy()
        rows.append({"max_iter": max_iter, "iter": it, "beta": float(reg.beta), "value": float(v)})
    meta.extend(rows)
    out["rect_sigmoid"] = aq.adaround_reg().rectified_sigmoid(mask).numpy()


def l2_cases(aq, out, meta):
    for key, shape in (("l2_conv", (4, 6, 5, 5)), ("l2_gemm", (7, 10))):
        g = torch.Generator().manual_seed(31 + len(shape))
        pred = torch.randn(shape, generator=g).requires_grad_(True)
        tgt = torch.randn(shape, generator=g)
        v = aq.L2_norm(pred, tgt)
        v.backward()
        out[key + "_pred"], out[key + "_tgt"], out[key + "_grad"] = pred.detach().numpy(), tgt.numpy(), pred.grad.numpy()
        meta.append({"key": key, "value": float(v)})


def acti_drop_case(aq, out, meta):
    """quant_acti with prob < 1: the random draw is prescribed (torch.rand_like patched) so it can be replayed."""
    g = torch.Generator().manual_seed(77)
    x = (torch.randn((3, 8, 6, 6), generator=g) * 2.0).requires_grad_(True)
    r = torch.rand((3, 8, 6, 6), generator=g)
    G = torch.randn((3, 8, 6, 6), generator=g)
    scale, q_min, q_max = torch.tensor(0.031), torch.tensor(-127.0), torch.tensor(127.0)
    with mock.patch.object(aq.torch, "rand_like", lambda t: r):
        y = aq.quant_acti(x, scale, q_min, q_max, 0.5)
    (y * G).sum().backward()
    out["drop_x"], out["drop_r"], out["drop_G"] = x.detach().numpy(), r.numpy(), G.numpy()
    out["drop_y"], out["drop_grad"] = y.detach().numpy(), x.grad.numpy()
    meta.append({"key": "drop", "scale": 0.031, "q_min": -127.0, "q_max": 127.0, "prob": 0.5})


def trajectories(aq, out, meta):
    """learning_round_mask (adaround.py:119-144) without DDP, CPU torch: a Gemm layer (per-channel scales) and
    a Conv + ReLU layer (per-tensor scale).  Stores the round mask after 1, 10 and all steps."""
    import torch.nn.functional as F
    for key, kind in (("traj_gemm", "gemm"), ("traj_conv", "conv")):
        g = torch.Generator().manual_seed(5 if kind == "gemm" else 6)
        if kind == "gemm":
            n, bs, epochs = 32, 16, 30
            w = torch.randn((8, 16), generator=g) * 0.2
            b = torch.randn((8,), generator=g) * 0.1
            x = torch.randn((n, 16), generator=g)
            scale = channel_view(w.abs().max(1).values / 127.0, 2)
            q_min, q_max = channel_view(torch.full((8,), -127.0), 2), channel_view(torch.full((8,), 127.0), 2)
            per_channel, relu = True, False
            fwd = lambda qw: F.linear(xb, qw, b)  # noqa: E731
            fp = F.linear(x + torch.randn(x.shape, generator=g) * 0.05, w, b)   # fp input != quantised input
        else:
            n, bs, epochs = 16, 8, 30
            w = torch.randn((6, 4, 3, 3), generator=g) * 0.1
            b = torch.randn((6,), generator=g) * 0.1
            x = torch.randn((n, 4, 8, 8), generator=g)
            scale = (w.abs().max() / 7.0).reshape(())       # a 4-bit-like grid: rounding matters
            q_min, q_max = torch.tensor(-7.0), torch.tensor(7.0)
            per_channel, relu = False, True
            fwd = lambda qw: F.relu(F.conv2d(xb, qw, b, 1, 1))  # noqa: E731
            fp = F.relu(F.conv2d(x + torch.randn(x.shape, generator=g) * 0.05, w, b, 1, 1))
        total_iter = epochs * int(np.ceil(n / bs))
        reg = aq.adaround_reg(total_iter)
        rest = (w / scale) - (w / scale).floor()
        mask = torch.nn.Parameter(-torch.log((reg.zeta - reg.gamma) / (rest - reg.gamma) - 1))
        opt = torch.optim.Adam([mask])
        cur, snaps, losses = 0, {}, []
        for ep in range(epochs):
            for idx in range(int(np.ceil(n / bs))):
                xb = x[idx * bs:(idx + 1) * bs]
                outp = fwd(aq.quant_weight(w, mask, scale, q_min, q_max, per_channel))
                loss = aq.L2_norm(outp, fp[idx * bs:(idx + 1) * bs]) + reg(mask, cur)
                cur += 1
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss))
                if cur in (1, 10, total_iter):
                    snaps[cur] = mask.detach().clone().numpy()
        hard = aq.quant_weight(w, mask.detach(), scale, q_min, q_max, per_channel, soft=False)
        out[key + "_w"], out[key + "_b"], out[key + "_x"] = w.numpy(), b.numpy(), x.numpy()
        out[key + "_fp"] = fp.numpy()
        out[key + "_scale"] = scale.reshape(-1).numpy()
        for k, v in snaps.items():
            out[f"{key}_mask_{k}"] = v
        out[key + "_hard"] = hard.numpy()
        out[key + "_losses"] = np.array(losses, np.float64)
        meta.append({"key": key, "kind": kind, "n": n, "bs": bs, "epochs": epochs, "total_iter": total_iter,
                     "per_channel": per_channel, "relu": relu, "q_min": float(q_min.reshape(-1)[0]),
                     "q_max": float(q_max.reshape(-1)[0])})


class _Init(np.ndarray):
    """An initializer stand-in: an array with the .name the reference reads (TensorProto.name)."""
    def __new__(cls, arr, name):
        o = np.asarray(arr).view(cls)
        o.name = name
        return o

    def __array_finalize__(self, obj):
        self.name = getattr(obj, "name", None)


class _FakeGraph:
    """What weight_equalization / update_bn_node touch of the reference's ONNXGraph."""
    def __init__(self, nodes=(), inits=None):
        import types
        self.graph = types.SimpleNamespace(node=list(nodes))
        self.initializer = {k: [_Init(v, k)] for k, v in (inits or {}).items()}
        self.saved = None

    def copy_from(self, g):
        import copy
        self.graph = copy.deepcopy(g.graph)
        self.initializer = {k: [_Init(np.array(v[0]), k)] for k, v in g.initializer.items()}

    def get_tensor_consumer(self, t):
        res = [n for n in self.graph.node if t in n.input]
        return res if res else ["OUTPUT_TOKEN"]

    def set_initializer(self, name, arr, raw=True):
        self.initializer[name] = [_Init(np.array(arr), name)]

    def update_model(self):
        pass

    def save_onnx_model(self, name):
        self.saved = name


def we_case(out, meta):
    """weight_equalization (weight_equalization.py:38-94) on a small chain: conv1 -> relu -> conv2 -> conv3(grouped)
    -> conv4, plus a branch that must be left alone."""
    import types
    import dipoorlet.weight_transform.weight_equalization as we
    we.numpy_helper.to_array = lambda t: np.asarray(t)
    holder = {}

    class G(_FakeGraph):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            holder["last"] = self
    we.ONNXGraph = G
    N = lambda op, i, o, name: types.SimpleNamespace(op_type=op, input=list(i), output=list(o), name=name)  # noqa: E731
    nodes = [N("Conv", ["x", "w1", "b1"], ["c1"], "conv1"), N("Relu", ["c1"], ["r1"], "relu1"),
             N("Conv", ["r1", "w2", "b2"], ["c2"], "conv2"), N("Conv", ["c2", "w3"], ["c3"], "conv3"),
             N("PRelu", ["c3", "slope"], ["p3"], "prelu3"), N("Conv", ["p3", "w4", "b4"], ["c4"], "conv4"),
             N("Conv", ["c4", "w5"], ["c5"], "conv5"), N("Add", ["c5", "c4"], ["y"], "add")]
    g = torch.Generator().manual_seed(91)
    rnd = lambda *s: (torch.randn(s, generator=g) * 0.3).numpy()  # noqa: E731
    inits = {"w1": rnd(8, 3, 3, 3), "b1": rnd(8), "w2": rnd(12, 8, 3, 3) * 4.0, "b2": rnd(12),
             "w3": rnd(12, 3, 3, 3), "w4": rnd(6, 12, 1, 1), "b4": rnd(6), "w5": rnd(6, 6, 3, 3), "slope": rnd(12)}
    inits["w1"][2] *= 1e-8          # a dead output channel: range below 1e-6 -> s := 1
    inits["w4"][:, 5] *= 30.0
    args = types.SimpleNamespace()
    we.weight_equalization(_FakeGraph(nodes, inits), args)
    res = holder["last"]
    assert res.saved == "weight_equal_model"
    for k, v in inits.items():
        out["we_in_" + k] = v
        out["we_out_" + k] = np.array(res.initializer[k][0])
    meta["we"] = {"nodes": [[n.op_type, n.input, n.output, n.name] for n in nodes]}


def bn_case(out, meta):
    """update_bn_node (update_bn.py:12-23)."""
    import types
    import dipoorlet.weight_transform.update_bn as ub
    ub.numpy_helper.to_array = lambda t: np.asarray(t)
    g = torch.Generator().manual_seed(17)
    xs = [(torch.randn((1, 5, 6, 7), generator=g) * (1.0 + 0.2 * i) + 0.3 * i).numpy() for i in range(9)]
    mean0 = torch.randn(5, generator=g).numpy()
    var0 = torch.rand(5, generator=g).numpy() + 0.5
    fg = _FakeGraph([], {"m": mean0, "v": var0})
    node = types.SimpleNamespace(input=["x", "scale", "bias", "m", "v"])
    ub.update_bn_node(fg, node, xs)
    out["bn_x"] = np.stack(xs)
    out["bn_mean0"], out["bn_var0"] = mean0, var0
    out["bn_mean1"], out["bn_var1"] = np.array(fg.initializer["m"][0]), np.array(fg.initializer["v"][0])
    meta["bn"] = {"dtype_mean": str(out["bn_mean1"].dtype), "dtype_var": str(out["bn_var1"].dtype)}


def sparse_cases(out, meta):
    """prune masks, quant_weight_wo_roundmask (value + straight-through gradient) and an SGD trajectory of
    learning_sparse_quant (sparse_quant.py:107-130) on CPU torch, through the reference's own functions."""
    import torch.nn.functional as F
    import dipoorlet.weight_transform.sparse_quant_layer as sq
    g = torch.Generator().manual_seed(123)
    w4 = torch.randn((8, 8, 3, 3), generator=g) * 0.1
    w2 = torch.randn((6, 16), generator=g) * 0.1
    out["sp_w4"], out["sp_w2"] = w4.numpy(), w2.numpy()
    out["sp_mask_unstr_w4"] = sq.create_unstruction_mask(w4, 0.5).numpy()
    out["sp_mask_unstr_w2_30"] = sq.create_unstruction_mask(w2, 0.3).numpy()
    out["sp_mask_nv24_w4"] = sq.create_nv24_mask(w4, 2, 4).numpy()
    out["sp_mask_nv24_w2"] = sq.create_nv24_mask(w2, 2, 4).numpy()
    rows = []
    for key, w, per_channel in (("spq_pc", w4, True), ("spq_pt", w2, False)):
        if per_channel:
            scale = channel_view(w.abs().reshape(w.shape[0], -1).max(1).values / 200.0, w.dim())   # some values clamp
            q_min, q_max = channel_view(torch.full((w.shape[0],), -127.0), w.dim()), channel_view(torch.full((w.shape[0],), 127.0), w.dim())
        else:
            scale, q_min, q_max = (w.abs().max() / 127.0).reshape(()), torch.tensor(-127.0), torch.tensor(127.0)
        G = torch.randn(w.shape, generator=g)
        wp = w.clone().requires_grad_(True)
        info = {"pattern": "unstruction", "rate": 0.5}
        qw = sq.quant_weight_wo_roundmask(sq.prune_weight(wp, info), scale, q_min, q_max, per_channel)
        (qw * G).sum().backward()
        out[key + "_scale"], out[key + "_G"] = scale.reshape(-1).numpy(), G.numpy()
        out[key + "_qw"], out[key + "_grad"] = qw.detach().numpy(), wp.grad.numpy()
        rows.append({"key": key, "per_channel": per_channel, "w": "sp_w4" if per_channel else "sp_w2"})
    meta["sparse_quant"] = rows
    # trajectory: conv 3x3 pad 1 + relu, per-channel grid, unstructured 50 %
    n, bs, epochs = 16, 8, 12
    w = torch.randn((6, 4, 3, 3), generator=g) * 0.1
    b = torch.randn((6,), generator=g) * 0.1
    x = torch.randn((n, 4, 8, 8), generator=g)
    fp = F.relu(F.conv2d(x + torch.randn(x.shape, generator=g) * 0.05, w, b, 1, 1))
    scale = channel_view(w.abs().reshape(6, -1).max(1).values / 127.0, 4)
    q_min, q_max = channel_view(torch.full((6,), -127.0), 4), channel_view(torch.full((6,), 127.0), 4)
    info = {"pattern": "unstruction", "rate": 0.5}
    wp = torch.nn.Parameter(w.clone())
    opt = torch.optim.SGD([wp], lr=0.001, momentum=0.9, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=epochs)
    losses = []
    for ep in range(epochs):
        for idx in range(n // bs):
            xb = x[idx * bs:(idx + 1) * bs]
            qw = sq.quant_weight_wo_roundmask(sq.prune_weight(wp, info), scale, q_min, q_max, True)
            loss = sq.L2_norm(F.relu(F.conv2d(xb, qw, b, 1, 1)), fp[idx * bs:(idx + 1) * bs])
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        sched.step()
    final = sq.quant_weight_wo_roundmask(sq.prune_weight(wp.detach(), info), scale, q_min, q_max, True)
    out["sptraj_w"], out["sptraj_b"], out["sptraj_x"], out["sptraj_fp"] = w.numpy(), b.numpy(), x.numpy(), fp.numpy()
    out["sptraj_scale"] = scale.reshape(-1).numpy()
    out["sptraj_learned"], out["sptraj_final"] = wp.detach().numpy(), final.numpy()
    out["sptraj_losses"] = np.array(losses)
    meta["sparse_traj"] = {"n": n, "bs": bs, "epochs": epochs, "rate": 0.5}


def main():
    aq = import_reference()
    torch.set_num_threads(1)
    out, meta = {}, {"torch": torch.__version__, "numpy": np.__version__, "quant_weight": [], "reg": [], "l2": [],
                     "drop": [], "traj": []}
    quant_weight_cases(aq, out, meta["quant_weight"])
    reg_cases(aq, out, meta["reg"])
    l2_cases(aq, out, meta["l2"])
    acti_drop_case(aq, out, meta["drop"])
    trajectories(aq, out, meta["traj"])
    we_case(out, meta)
    sparse_cases(out, meta)
    bn_case(out, meta)
    t = aq.TempDecay(1000)
    meta["temp_decay_1000"] = {str(i): float(t(i)) for i in (0, 199, 200, 500, 1000)}
    np.savez_compressed(os.path.join(HERE, "round_level.npz"), **out)
    with open(os.path.join(HERE, "round_level.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote round_level.{npz,json}:", len(out), "arrays")


if __name__ == "__main__":
    main()
