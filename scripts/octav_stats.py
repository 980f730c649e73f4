#!/usr/bin/env python3
"""Tuning helper (GPU box): how many values the bracket form gathers and how many pairs fall back, on the bench's
synthetic ResNet-50 activations."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipoorlet_amd import _hip, ops  # noqa: E402
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations  # noqa: E402

dev = torch.device("cuda:0")
spec = resnet50_tensors()
elems = [e for _, e, _ in spec]
B = 16
plan = ops.TensorSetPlan(elems, B, dev)
t = synth_activations(spec, B, dev, seed=1234)
states = torch.empty((plan.n_pairs + 1) * ctypes.sizeof(_hip.OctavState), dtype=torch.uint8, device=dev)
rows = ops.octav_batch(plan, t, False, states)
torch.cuda.synchronize()
raw = states.cpu().numpy().view(np.dtype([("sum", "<f8"), ("cnt_gt", "<u8"), ("cnt_le", "<u8"), ("min", "<u4"), ("max", "<u4"),
                                          ("nan", "<u4"), ("done", "<u4"), ("s", "<f4"), ("ud", "<f4"), ("iters", "<u4"),
                                          ("mode", "<u4"), ("n", "<u8"), ("len0", "<u4"), ("len1", "<u4"), ("cur", "<u4"),
                                          ("res", "<u4")]))
pairs, ctl = raw[:-1], raw[-1]
print("pairs", len(pairs), "elements", int(pairs["n"].sum()), "gathered", int(pairs["len0"].sum()),
      "fraction %.4f" % (pairs["len0"].sum() / pairs["n"].sum()), "fallback pairs (compaction route)", int(ctl["cnt_le"]),
      "mode histogram", np.bincount(pairs["mode"]).tolist(), "checksum %.6f" % float(rows[..., 0].double().sum()))
