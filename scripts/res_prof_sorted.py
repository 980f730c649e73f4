#!/usr/bin/env python3
"""Phase breakdown of k_octav_walk_sorted (a -DDPL_RES_PROF build): clock ticks per phase and pair, by pair size class."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dipoorlet_amd import _hip, ops  # noqa: E402
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations  # noqa: E402

B = 32
dev = torch.device("cuda:0")
spec = resnet50_tensors()
elems = [e for _, e, _ in spec]
pool = [synth_activations(spec, B, dev, seed=1234 + j) for j in range(4)]
plan = ops.TensorSetPlan(elems, B, dev)
L = _hip.lib()
L.dpl_res_prof_read.restype = C.c_int
L.dpl_res_prof_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(4096 * 8, np.uint64)
for i in range(6):
    ops.octav_batch(plan, pool[i % 4], False, form="oneread")
torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 1)
n = 8
for i in range(n):
    ops.octav_batch(plan, pool[i % 4], False, form="oneread")
torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 0)
p = buf.reshape(4096, 8).astype(np.float64) / n
for lo, hi in ((0, 416), (416, 896), (896, 1952), (1952, 3840)):
    q = p[lo:hi]
    print(f"pairs {lo:4d}-{hi:4d}: rows {q[:, 0].mean():8.0f}  walk {q[:, 2].mean():8.0f} ticks  iterations {q[:, 7].mean():.1f}  -> {q[:, 2].mean() / max(1e-9, q[:, 7].mean()):.0f} ticks per iteration")
