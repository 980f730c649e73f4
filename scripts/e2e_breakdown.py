#!/usr/bin/env python3
"""Where does the end-to-end time go? (GPU box)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dipoorlet_amd import models, ops
from dipoorlet_amd.forward_net import load_input_batch

N, B = 1024, 32
d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, "input"))
rng = np.random.default_rng(0)
blob = rng.standard_normal(3 * 224 * 224).astype(np.float32)
for i in range(N):
    blob.tofile(os.path.join(d, "input", f"{i}.bin"))
dev = torch.device("cuda:0")
t = time.time(); g = models.resnet50(); print("build graph %.2f" % (time.time() - t))
t = time.time(); s = g.make_session(); torch.cuda.synchronize(); print("session %.2f" % (time.time() - t))
shapes = {"input": g.get_tensor_shape("input")}
t = time.time()
ins = [load_input_batch(d, ["input"], shapes, i, i + B, dev) for i in range(0, N, B)]
torch.cuda.synchronize(); print("load .bin %.2f s (%.0f img/s)" % (time.time() - t, N / (time.time() - t)))
for rep in range(2):
    t = time.time()
    outs = [s.run(x) for x in ins[:8]]
    torch.cuda.synchronize(); dt = time.time() - t
    print("forward rep%d: %.3f s for %d imgs (%.0f img/s)" % (rep, dt, 8 * B, 8 * B / dt))
    del outs
plan = ops.TensorSetPlan(s.elems_per_image, B, dev)
acc = ops.CalibAccumulators(len(s.elems_per_image), dev)
o = s.run(ins[0]); torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    acc.minmax_accumulate(plan, o)
torch.cuda.synchronize(); print("minmax per batch %.2f ms" % ((time.time() - t) * 100))
