#!/usr/bin/env python3
"""Phase clock counters of k_octav_probe (a -DDPL_RES_PROF build): sample loads + LDS atomics / conversion + scans / the
sample's walk by one thread, per workgroup (workgroups are in largest-pair-first order)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dipoorlet_amd import _hip, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
os.environ["DPL_OCTAV_PREDICT"] = "probe"
os.environ["DPL_OCTAV_FUSE"] = "0"        # (the fused kernel's own counters share the slots)
dev = torch.device("cuda:0")
spec = resnet50_tensors(); elems = [e for _, e, _ in spec]; B = 32
x = synth_activations(spec, B, dev, seed=3)
plan = ops.TensorSetPlan(elems, B, dev)
L = _hip.lib()
L.dpl_res_prof_read.restype = C.c_int
L.dpl_res_prof_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(4096 * 8, np.uint64)
ops.octav_batch(plan, x, False); torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 1)
res = plan.octav_oneread_scratch()
job = ops._oneread_job(plan, res, plan.seg_table(x), torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev), res["lh"], res["pred"],
                       res["pred_pair"], res["use_probe"], plan.octav_scratch()[3], 0, 0, 0, 0)
res["use_probe"].fill_(1)
n = 10
for _ in range(n):
    _hip.check(L.dpl_octav_oneread_probe(C.byref(job), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "probe")
torch.cuda.synchronize()
L.dpl_res_prof_read(buf.ctypes.data, 0)
p = buf.reshape(4096, 8).astype(np.float64) / n
for lo, hi in ((0, 384), (384, 864), (864, 1952), (1952, 3936)):
    q = p[lo:hi]
    print(f"workgroups {lo:4d}-{hi:4d}: loads+atomics {q[:, 0].mean():8.0f}  convert+scan {q[:, 1].mean():8.0f}  walk {q[:, 2].mean():8.0f} ticks")
print("sum over workgroups: loads %.3e convert %.3e walk %.3e" % (p[:, 0].sum(), p[:, 1].sum(), p[:, 2].sum()))
