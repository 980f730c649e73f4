"""GPU diagnostic: which (image, tensor) pairs of a batch the one-read OCTAV form rescues / sends to the compaction route,
by tensor.  python scripts/octav_rescue_stats.py [resnet50|vit] [batches]"""
import os
import sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dipoorlet_amd import _hip, models, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations

dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if which == "vit":
    g = models.vit_b16(seed=5, attn_gain=10.0)
    sess = g.make_session()
    names, elems, B = list(sess.tensor_names), [int(e) for e in sess.elems_per_image], 8
    prod = {n.output[0]: n.op_type for n in g.graph.node}
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    batch = lambda k: [t.reshape(B, -1) for t in sess.run({"input": torch.randn(B, 3, 224, 224, generator=gen, device=dev)})]
else:
    spec = resnet50_tensors()
    names, elems, B = [n for n, _, _ in spec], [e for _, e, _ in spec], 32
    prod = {n: k for n, _, k in spec}
    batch = lambda k: synth_activations(spec, B, dev, seed=500 + k)
T = len(elems)
plan = ops.TensorSetPlan(elems, B, dev)
states = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
for k in range(nb):
    x = batch(k)
    ops.octav_batch(plan, x, False, states)
    torch.cuda.synchronize()
    raw = states.cpu().numpy().reshape(-1, 80)
    ctl = _hip.OctavState.from_buffer_copy(raw[-1].tobytes())
    res = plan.octav_oneread_scratch()
    missed = res["missed"].cpu().numpy()[:ctl.len0, 0]
    mode = raw[:-1, 52:56].copy().view(np.uint32).reshape(-1)      # dpl_octav_state.mode
    use = res["use_probe"].cpu().numpy()
    print(f"batch {k}: rescued {ctl.len0} units {ctl.len1} compaction {ctl.cnt_le} listed {ctl.sum / (B * sum(elems)):.4f} "
          f"tiles twice {ctl.reserved} own-sample tensors {int(use.sum())}/{T}", flush=True)
    if k == nb - 1:
        by = Counter((prod.get(names[p % T], "input"), elems[p % T]) for p in missed)
        print("rescued by (producer, elems):", sorted(by.items(), key=lambda kv: -kv[1])[:25])
        iters = raw[:-1, 48:52].copy().view(np.uint32).reshape(-1)
        comp = [p for p in range(plan.n_pairs) if mode[p] in (0, 1)]
        print("compaction route:", [(names[p % T], prod.get(names[p % T], "input"), elems[p % T], int(iters[p]),
                                     float(x[p % T][p // T].abs().max()), float(x[p % T][p // T].abs().min())) for p in comp][:12])
