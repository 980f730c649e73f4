"""GPU workload for a kernel trace: the fake-quantised ResNet-50 forward of bench.py's `fake_quant.product_forward` (quant_graph for -D
trt, batch 64) in its two forms, 3 warm-up + 10 counted forwards each, a k_channel_diff_sum launch as delimiter in front of each counted
block:
  A  every tensor exposed (GraphSession.run: ReLU, Add and Q/DQ are separate launches — what profiling's per-layer pass needs)
  B  only the network output asked for (run_named): a ReLU / Add + ReLU whose only reader is a Q/DQ pair runs inside k_fake_quant<PRE>
     (executor.relu_fusion — what --bc, update_bn and AdaRound's walks and any caller of the quantised network's output run)
Prints the Q/DQ nodes and the bytes they move per forward in each form (8 B per element; 12 B for Add + ReLU + Q/DQ: two reads).
rocprofv3 --kernel-trace -- python3 scripts/fq_forward_run.py   (scripts/fq_forward_prof.sh)"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import executor as ex, models, ops
from dipoorlet_amd.quantize import quant_graph
from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
dev = torch.device("cuda")
B = int(os.environ.get("DPL_FQ_BATCH", "64"))
g = models.resnet50()
s = g.make_session()
gen = torch.Generator(device=dev); gen.manual_seed(99)
x = torch.randn(B, 3, 224, 224, generator=gen, device=dev)
clip = {n: [float(t.amin()), float(t.amax())] for n, t in zip(s.tensor_names, s.run({"input": x}))}
clip.update(find_clip_val_minmax_weight(g, None, session=s))
del s
gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
sq = gq.make_session()
out_name = gq.network_outputs[0]
count = {"A": [0, 0], "B": [0, 0]}
mode = [None]
orig, orig_fused = ex._OPS["FakeQuant"], ex.fused_fake_quant


def counted(sess, node, t):
    if mode[0]:
        count[mode[0]][0] += 1
        count[mode[0]][1] += 8 * t.numel()
    return orig(sess, node, t)


def counted_fused(sess, node, pre, *xs):
    if mode[0]:
        count[mode[0]][0] += 1
        count[mode[0]][1] += (12 if pre == "add_relu" else 8) * xs[0].numel()
    return orig_fused(sess, node, pre, *xs)


ex._OPS["FakeQuant"], ex.fused_fake_quant = counted, counted_fused
delim_a, delim_b = torch.ones(2, 4, 8, device=dev), torch.zeros(2, 4, 8, device=dev)
N = 10
res = {}
for tag, fwd in (("A", lambda: sq.run({"input": x})), ("B", lambda: sq.run_named({"input": x}, [out_name]))):
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    ops.channel_diff_sum(delim_a, delim_b)
    mode[0] = tag
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    mode[0] = None
    res[tag] = {"forwards": N, "qdq_nodes_per_forward": count[tag][0] // N, "qdq_bytes_per_forward": count[tag][1] // N,
                "forward_ms_by_events": e0.elapsed_time(e1) / N}
ops.channel_diff_sum(delim_a, delim_b)
torch.cuda.synchronize()
fused, skipped = sq.fusion([out_name])
res["fused_pairs"] = {"relu": sum(1 for p, _ in fused.values() if p == "relu"), "add_relu": sum(1 for p, _ in fused.values() if p == "add_relu"),
                      "launches_saved_per_forward": len(skipped)}
res["batch"] = B
ex.join_helpers()
print("FQFWD " + json.dumps(res))
