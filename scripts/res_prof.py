#!/usr/bin/env python3
"""Phase breakdown of the register-resident OCTAV kernel (a -DDPL_RES_PROF build, see scripts/build_variant.sh):
cycles per phase summed over workgroups, as a share of (workgroups x kernel time)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dipoorlet_amd import _hip, ops  # noqa: E402
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
spec = resnet50_tensors()
elems = [e for _, e, _ in spec]
pool = [synth_activations(spec, B, dev, seed=1234 + j) for j in range(4)]
plan = ops.TensorSetPlan(elems, B, dev)
L = _hip.lib()
L.dpl_res_prof_read.restype = C.c_int
L.dpl_res_prof_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(4096 * 8, np.uint64)
for i in range(3):
    ops.octav_batch(plan, pool[i % 4], False, form="oneread")
torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 8
e0.record()
for i in range(n):
    ops.octav_batch(plan, pool[i % 4], False, form="oneread")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
L.dpl_res_prof_read(buf.ctypes.data, 0)
st = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
ops.octav_batch(plan, pool[0], False, st, form="oneread")
print("pairs on the compaction route in a steady-state batch:", int(st.cpu().numpy()[-80:].view(np.uint64)[2]), "of", plan.n_pairs)
p = buf.reshape(4096, 8).astype(np.float64)
wgs = int((p.sum(1) > 0).sum())
names = ["rows -> suffix totals, s_0", "list -> registers", "walk", "publish"]
tot = p.sum(0) / n
wgs = int((p[:, 0] > 0).sum())
print(f"batch {B}: {ms:.3f} ms per batch (all kernels); walk kernel: {wgs} workgroups, mean iterations {tot[7] / wgs:.1f}, "
      f"mean list {tot[6] / wgs:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {tot[i] / wgs:9.0f} ticks per workgroup   max {p[:, i].max() / n:9.0f}")
for lo, hi in ((0, 384), (384, 864), (864, 1952), (1952, 3936)):
    q = p[lo:hi] / n
    print(f"  slots {lo:4d}-{hi:4d}: rows {q[:, 0].mean():8.0f}  list {q[:, 1].mean():8.0f}  walk {q[:, 2].mean():8.0f}  publish {q[:, 3].mean():8.0f}  "
          f"iters {q[:, 7].mean():.1f}  list {q[:, 6].mean():.0f}   | stream {q[:, 4].mean():9.0f}  fused walk total {q[:, 5].mean():8.0f}")
print(f"  sums over workgroups (ticks): stream {p[:, 4].sum() / n:.3e}  fused walk {p[:, 5].sum() / n:.3e}  -> walk share of workgroup time "
      f"{p[:, 5].sum() / max(1.0, p[:, 4].sum() + p[:, 5].sum()):.3f}")
