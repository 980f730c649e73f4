"""GPU workload for a kernel trace: the fake-quantised ResNet-50 forward of bench.py's `fake_quant.product_forward` (quant_graph for -D
trt, batch 64), 3 warm-up forwards + 10 counted ones; prints the bytes its 55 activation Q/DQ nodes move per forward.
rocprofv3 --kernel-trace --stats -- python3 scripts/fq_forward_run.py   (scripts/fq_forward_prof.sh)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import executor as ex, models
from dipoorlet_amd.quantize import quant_graph
from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
dev = torch.device("cuda")
g = models.resnet50()
s = g.make_session()
gen = torch.Generator(device=dev); gen.manual_seed(99)
x = torch.randn(64, 3, 224, 224, generator=gen, device=dev)
clip = {n: [float(t.amin()), float(t.amax())] for n, t in zip(s.tensor_names, s.run({"input": x}))}
clip.update(find_clip_val_minmax_weight(g, None, session=s))
del s
gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
sq = gq.make_session()
count = [0, 0]
orig = ex._OPS["FakeQuant"]
def counted(sess, node, t):
    count[0] += 1; count[1] += t.numel()
    return orig(sess, node, t)
for _ in range(3):
    sq.run({"input": x})
ex._OPS["FakeQuant"] = counted
N = 10
for _ in range(N):
    sq.run({"input": x})
torch.cuda.synchronize()
ex.join_helpers()
print("forwards %d, Q/DQ nodes per forward %d, bytes per forward %d" % (N, count[0] // N, 8 * count[1] // N))
