#!/usr/bin/env python3
"""AdaRound inner-loop timing on one MI355X: the fused path (dipoorlet_amd.weight_transform) against the same
arithmetic written as eager torch ops with autograd + torch.optim.Adam (what the reference runs), on ResNet-50
layer shapes.  Prints one JSON line per layer:  python scripts/round_bench.py [--iters 200]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [  # name, Cin, Cout, k, H, stride, pad
    ("layer1.conv2 3x3 64->64 @56", 64, 64, 3, 56, 1, 1),
    ("layer1.conv3 1x1 64->256 @56", 64, 256, 1, 56, 1, 0),
    ("layer3.conv2 3x3 256->256 @14", 256, 256, 3, 14, 1, 1),
    ("layer4.conv2 3x3 512->512 @7", 512, 512, 3, 7, 1, 1),
]


def eager_loop(w, b, x, fp, scale, iters, total, pad, stride):
    """ada_quant_layer.py / adaround.py semantics in eager torch (per-channel clamp, ReLU, L2 + regulariser, Adam)."""
    zeta, gamma, lam = 1.1, -0.1, 0.01
    s = scale.reshape(-1, 1, 1, 1)
    rest = (w / s) - (w / s).floor()
    mask = torch.nn.Parameter(-torch.log((zeta - gamma) / (rest - gamma) - 1))
    opt = torch.optim.Adam([mask])
    qmin, qmax = torch.full_like(s, -127.0), torch.full_like(s, 127.0)

    def h(m):
        return ((zeta - gamma) * torch.sigmoid(m) + gamma).clamp(0, 1)
    bs = x.shape[0] // 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        st = (it % 2) * bs
        qw = torch.min(torch.max((w / s).floor() + h(mask), qmin), qmax) * s
        out = F.relu(F.conv2d(x[st:st + bs], qw, b, stride, pad))
        beta = 20.0 if it >= 0 else 0.0
        loss = (out - fp[st:st + bs]).pow(2.0).sum(1).mean() + lam * (1 - torch.pow((h(mask) - 0.5).abs() * 2, beta)).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def fused_loop(w, b, x, fp, scale, iters, total, pad, stride, k):
    from dipoorlet_amd.onnx_io import Node
    from dipoorlet_amd.weight_transform.ada_quant_layer import AdaQLayer, L2_norm
    node = Node("Conv", ["x", "w", "b"], ["y"], name="c", attrs={"pads": [pad] * 4, "kernel_shape": [k, k],
                                                                  "strides": [stride] * 2, "dilations": [1, 1], "group": 1})
    qw = {"scale": scale, "q_min": torch.full_like(scale, -127.0), "q_max": torch.full_like(scale, 127.0),
          "per_channel": True, "type": "Linear"}
    layer = AdaQLayer(node, w, b, qw, None, True, False)
    loss = torch.zeros(2, dtype=torch.float64, device=w.device)
    bs = x.shape[0] // 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        st = (it % 2) * bs
        z = layer(x[st:st + bs], apply_relu=False)
        loss.zero_()
        _, grad = L2_norm(z, fp[st:st + bs], relu=True, loss=loss[0:1])
        z.backward(grad)
        layer.rp.step(20.0, reg_loss=loss[1:2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def learner_loop(w, b, x, fp, scale, iters, pad, stride, k, use_graph):
    """The product's learn_rounding (eager launches or hipGraph replay) on the same layer: ms per iteration."""
    from dipoorlet_amd.onnx_io import Node
    from dipoorlet_amd.weight_transform.ada_quant_layer import AdaQLayer, adaround_reg
    from dipoorlet_amd.weight_transform.reconstruction import learn_rounding
    node = Node("Conv", ["x", "w", "b"], ["y"], name="c", attrs={"pads": [pad] * 4, "kernel_shape": [k, k],
                                                                  "strides": [stride] * 2, "dilations": [1, 1], "group": 1})
    qw = {"scale": scale, "q_min": torch.full_like(scale, -127.0), "q_max": torch.full_like(scale, 127.0),
          "per_channel": True, "type": "Linear"}
    layer = AdaQLayer(node, w, b, qw, None, True, False)
    epochs = max(2, iters // 2)
    fp_r = fp  # already ReLU'd
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    learn_rounding([layer], x, None, fp_r, adaround_reg(2 * epochs), x.shape[0] // 2, epochs, log_every=10 ** 9,
                   use_graph=use_graph)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * epochs)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iters", type=int, default=200)
    p.add_argument("--bs", type=int, default=64)
    a = p.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, cin, cout, k, hw, stride, pad in LAYERS:
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev) * 0.1
        x = torch.randn(2 * a.bs, cin, hw, hw, device=dev)
        fp = F.relu(F.conv2d(x + 0.05 * torch.randn_like(x), w, b, stride, pad))
        scale = w.abs().reshape(cout, -1).max(1).values / 127.0
        res = {}
        for label, fn in (("eager", lambda n: eager_loop(w, b, x, fp, scale, n, 0, pad, stride)),
                          ("fused", lambda n: fused_loop(w, b, x, fp, scale, n, 0, pad, stride, k))):
            fn(10)
            res[label + "_ms"] = round(fn(a.iters) * 1e3, 4)
        for label, g in (("learner_eager", False), ("learner_graph", True)):
            learner_loop(w, b, x, fp, scale, 20, pad, stride, k, g)
            res[label + "_ms"] = round(learner_loop(w, b, x, fp, scale, a.iters * 2, pad, stride, k, g) * 1e3, 4)
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                F.conv2d(x[:a.bs], w, b, stride, pad)
            torch.cuda.synchronize()
            res["conv_fwd_only_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 4)
        res.update(layer=name, batch=a.bs, speedup=round(res["eager_ms"] / res["fused_ms"], 2))
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
