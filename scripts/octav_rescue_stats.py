"""GPU debug: which pairs of a steady-state batch are rescued / end on the compaction route (one-read OCTAV)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dipoorlet_amd import _hip, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
dev = torch.device("cuda")
spec = resnet50_tensors(); elems = [e for _, e, _ in spec]; T = len(elems); B = 32
plan = ops.TensorSetPlan(elems, B, dev)
states = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
for k in range(12):
    x = synth_activations(spec, B, dev, seed=500 + k)
    out = ops.octav_batch(plan, x, False, states)
    torch.cuda.synchronize()
    raw = states.cpu().numpy().reshape(-1, 80)
    ctl = _hip.OctavState.from_buffer_copy(raw[-1].tobytes())
    res = plan.octav_oneread_scratch()
    missed = res["missed"].cpu().numpy()[:ctl.len0, 0]
    print(k, "rescued", ctl.len0, "units", ctl.len1, "compaction", ctl.cnt_le, "listed %.4f" % (ctl.sum / (B * sum(elems))),
          "rescued sizes", sorted(set(elems[p % T] for p in missed))[:8], flush=True)
