#!/usr/bin/env python3
"""Achieved HBM bandwidth of the auxiliary kernels (fake-quant, cosine sums, weight row ranges, L2 loss) on one
MI355X:  python scripts/aux_kernels_bench.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipoorlet_amd import ops  # noqa: E402
from dipoorlet_amd.weight_transform.ada_quant_layer import L2_norm  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


n = 32 * 256 * 56 * 56                       # one batch-32 ResNet-50 layer1 output: 25.7 M elements, 103 MB
x = torch.randn(32, 256, 56, 56, device=dev)
y = torch.randn_like(x)
out = torch.empty_like(x)
sc1, zp1 = torch.tensor([0.05], device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
scc, zpc = torch.rand(256, device=dev) * 0.1 + 0.01, torch.zeros(256, dtype=torch.int32, device=dev)
acc = torch.zeros(4, 3, dtype=torch.float64, device=dev)
w = torch.randn(2048, 512 * 9, device=dev)
loss = torch.zeros(1, dtype=torch.float64, device=dev)
rows = []
for name, fn, nbytes in (
        ("fake_quant per-tensor", lambda: ops.fake_quant(x, sc1, zp1, -128, 127, out=out), 8 * n),
        ("fake_quant per-channel (axis 1)", lambda: ops.fake_quant(x, scc, zpc, -128, 127, axis=1, out=out), 8 * n),
        ("cos_accumulate", lambda: ops.cos_accumulate(x, y, acc, 0), 8 * n),
        ("rowwise_minmax [2048, 4608]", lambda: ops.rowwise_minmax(w), 4 * w.numel()),
        ("L2_norm + grad (relu)", lambda: L2_norm(x, y, relu=True, loss=loss), 12 * n),
        ("torch: (x - y).pow(2).sum() reference pass", lambda: (x - y).pow(2).sum(), 8 * n)):
    t = timeit(fn)
    rows.append({"kernel": name, "ms": round(t * 1e3, 4), "GBps": round(nbytes / t / 1e9, 1), "of_8TBps": round(nbytes / t / 8e12, 3)})
    print(json.dumps(rows[-1]), flush=True)

# the batched cosine kernel of the profiling flow: every (image, tensor) pair of two ResNet-50-shaped tensor sets
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations  # noqa: E402
spec = resnet50_tensors()
elems = [e for _, e, _ in spec]
B = 16
plan = ops.TensorSetPlan(elems, B, dev)
ta, tb = synth_activations(spec, B, dev, seed=1), synth_activations(spec, B, dev, seed=2)
t = timeit(lambda: ops.cos_per_image(plan, ta, tb), iters=10)
nbytes = 8 * sum(elems) * B
print(json.dumps({"kernel": "cos_per_image (k_cos_items), ResNet-50 shapes, batch 16", "ms": round(t * 1e3, 4),
                  "GBps": round(nbytes / t / 1e9, 1), "of_8TBps": round(nbytes / t / 8e12, 3)}), flush=True)
