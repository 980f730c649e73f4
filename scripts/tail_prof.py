#!/usr/bin/env python3
"""Phase breakdown of k_octav_tail (a -DDPL_RES_PROF build of the library, DPL_LIB=...): ticks per phase and workgroup, by
size class of the pairs (the slices are launched largest first).  python scripts/tail_prof.py [jitter]"""
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
jit = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda:0")
spec = resnet50_tensors()
elems = [e for _, e, _ in spec]
pool = [synth_activations(spec, B, dev, seed=1234 + j, image_jitter=jit) for j in range(6)]
plan = ops.TensorSetPlan(elems, B, dev)
L = _hip.lib()
L.dpl_res_prof_read.restype = C.c_int
L.dpl_res_prof_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(4096 * 8, np.uint64)
for i in range(4):
    ops.octav_batch(plan, pool[i % 6], False, form="tail")
torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 12
e0.record()
for i in range(n):
    ops.octav_batch(plan, pool[(4 + i) % 6], False, form="tail")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
L.dpl_res_prof_read(buf.ctypes.data, 0)
p = buf.reshape(4096, 8).astype(np.float64) / n
print(f"jitter {jit}: {ms:.3f} ms per batch (one stream, all kernels)")
sizes = np.sort(np.array([e for _ in range(B) for e in elems]))[::-1]
edges = [0] + [int(i) + 1 for i in np.nonzero(np.diff(sizes))[0]] + [len(sizes)]
print("   pairs of  count |   stream    setup     bulk    exact  publish | walk total | evals  list")
for lo, hi in zip(edges[:-1], edges[1:]):
    q = p[lo:hi]
    print(f"  {sizes[lo]:9d} {hi - lo:6d} | {q[:, 4].mean():8.0f} {q[:, 0].mean():8.0f} {q[:, 1].mean():8.0f} {q[:, 2].mean():8.0f} {q[:, 3].mean():8.0f} | "
          f"{q[:, 5].mean():10.0f} | {q[:, 7].mean():5.1f} {q[:, 6].mean():6.0f}")
print(f"  sums over workgroups (ticks): stream {p[:, 4].sum():.3e}  walk {p[:, 5].sum():.3e}  -> walk share of workgroup time "
      f"{p[:, 5].sum() / max(1.0, p[:, 4].sum() + p[:, 5].sum()):.3f}")
