"""What makes two forwards of one session differ, and does MIOpen's naive convolution ever run in a user's process?
  One process, no helper threads (DPL_PREWARM_CONVS=0, warm_libraries not started):

  part A  every distinct convolution configuration of the network at `--batch` images, called `--calls` times on the SAME random
          input: HIP-event time of every call (a 5 ms outlier = a fallback kernel) and bit-equality of call k against call 1
  part B  `--forwards` forwards of ONE session over the same input: per forward the wall time, and for every exposed tensor
          bit-equality against forward 1; the ROOT differences are named (a node whose inputs are all bit-equal and whose output
          is not)

Under `rocprofv3 --kernel-trace -- python3 scripts/conv_repro_probe.py ...` every call of part A is preceded by ONE launch of this
library's k_gemm_small (two before a new configuration): scripts/conv_repro_kernels.py cuts the trace there and prints the kernel
names per (configuration, call).  --det: torch.backends.cudnn.deterministic = True (what DPL_DETERMINISTIC=1 sets)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DPL_PREWARM_CONVS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

p = argparse.ArgumentParser()
p.add_argument("--net", default="resnet50")
p.add_argument("--image", type=int, default=224)
p.add_argument("--batch", type=int, default=32)
p.add_argument("--calls", type=int, default=5)
p.add_argument("--forwards", type=int, default=4)
p.add_argument("--det", action="store_true")
p.add_argument("--quant", action="store_true", help="part B on the fake-quantised graph as well")
p.add_argument("--out", default="")
a = p.parse_args()

from dipoorlet_amd import executor as ex, models, ops

ex.warm_libraries = lambda *k, **kw: None          # no helper threads in this process
torch.backends.cudnn.deterministic = a.det
dev = torch.device("cuda")
g = getattr(models, a.net)(image=a.image) if a.net.startswith("resnet") else getattr(models, a.net)(seed=5, attn_gain=10.0)
s = g.make_session()
rec = {"net": a.net, "batch": a.batch, "det": a.det, "torch": torch.__version__, "miopen": torch.backends.cudnn.version(),
       "env": {k: v for k, v in os.environ.items() if k.startswith(("MIOPEN", "DPL_"))}}
tiny = torch.ones(4, 4, device=dev)


def delim(n):
    for _ in range(n):
        ops.gemm_small(tiny, tiny)


# ---------------------------------------------------------------- part A
gen = torch.Generator(device=dev)
gen.manual_seed(7)
seen, cfgs = set(), []
for node in g.graph.node:
    if node.op_type != "Conv":
        continue
    shp, w = s.shape1[node.input[0]], s._any_shape(node.input[1], 1)
    key = (tuple(shp[1:]), tuple(w), tuple(node.attrs.get("strides", ())), tuple(node.attrs.get("pads", ())), int(node.attrs.get("group", 1)),
           len(node.input))
    if key in seen:
        continue
    seen.add(key)
    cfgs.append((node, key))
A = []
for ci, (node, key) in enumerate(cfgs):
    x = torch.randn((a.batch,) + key[0], generator=gen, device=dev)
    w = torch.randn(key[1], generator=gen, device=dev) * 0.05
    b = torch.randn(key[1][0], generator=gen, device=dev) if key[5] > 2 else None
    torch.cuda.synchronize()
    delim(1)
    ms, same, first = [], [], None
    for k in range(a.calls):
        delim(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        y = ex._OPS["Conv"](s, node, x, w, b) if b is not None else ex._OPS["Conv"](s, node, x, w)
        e1.record()
        torch.cuda.synchronize()
        host_ms = 1e3 * (time.perf_counter() - t0)
        ms.append((round(e0.elapsed_time(e1), 3), round(host_ms, 1)))
        if first is None:
            first = y.clone()
        else:
            same.append(bool(torch.equal(first, y)))
    A.append({"i": ci, "node": node.name, "x": list(key[0]), "w": list(key[1]), "stride": list(key[2]), "pads": list(key[3]), "group": key[4],
              "gpu_ms_host_ms": ms, "bit_equal_to_call1": same, "maxdiff": float((first - y).abs().max())})
    print(json.dumps(A[-1]), flush=True)
rec["part_a"] = A
rec["part_a_not_reproducible"] = [r["node"] for r in A if not all(r["bit_equal_to_call1"])]
delim(3)

# ---------------------------------------------------------------- part B


def part_b(sess, graph, tag):
    x = {n: torch.randn([a.batch] + [int(d) for d in graph.get_tensor_shape(n)[1:]], generator=gen, device=dev) for n in sess.input_names}
    outs, walls = [], []
    for f in range(a.forwards):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = sess.run(x)
        torch.cuda.synchronize()
        walls.append(round(1e3 * (time.perf_counter() - t0), 2))
        outs.append([t.clone() for t in o] if f == 0 else [bool(torch.equal(u, v)) for u, v in zip(outs[0], o)] +
                    [float(max(float((u - v).abs().max()) for u, v in zip(outs[0], o)))])
    producer = {o: n for n in graph.graph.node for o in n.output}
    res = {"forward_ms": walls, "forwards": []}
    for f in range(1, a.forwards):
        eq = dict(zip(sess.tensor_names, outs[f][:-1]))
        diff = [n for n in sess.tensor_names if not eq[n]]
        roots = []
        for n in diff:
            node = producer.get(n)
            if node is None:
                continue
            if all(eq.get(i, True) for i in node.input):
                roots.append((node.op_type, node.name))
        res["forwards"].append({"vs_forward_1": f + 1, "tensors": len(eq), "differ": len(diff), "maxdiff": outs[f][-1], "root_nodes": roots[:40]})
    print(tag, json.dumps(res), flush=True)
    return res


rec["part_b_fp"] = part_b(s, g, "B fp")
if a.quant:
    import types
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
    clip = {n: [-3.0, 3.0] for n in s.tensor_names}
    clip.update(find_clip_val_minmax_weight(g, None, session=s))
    gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
    q1 = ex.GraphSession(gq, expose_fake_quant=True)
    rec["part_b_quant"] = part_b(q1, gq, "B quant")
    # two sessions of one graph
    q2 = ex.GraphSession(gq, expose_fake_quant=True)
    x = {n: torch.randn([a.batch] + [int(d) for d in g.get_tensor_shape(n)[1:]], generator=gen, device=dev) for n in q1.input_names}
    o1, o2 = q1.run(x), q2.run(x)
    rec["two_sessions_bit_equal"] = bool(all(torch.equal(u, v) for u, v in zip(o1, o2)))
    print("two quantised sessions bit-equal:", rec["two_sessions_bit_equal"], flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
