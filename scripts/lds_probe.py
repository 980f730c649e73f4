"""GPU: how the one-read OCTAV batch time depends on WHERE the values fall in the LDS histogram (same bytes, same shapes):
normal, log-uniform over the window (few same-bin collisions in a wave), one bin (every lane on one address), zeros (the lanes'
dummy words).  Tells an LDS-atomic bound from an HBM / issue bound.  python scripts/lds_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops

dev = torch.device("cuda")
B, T, E = 16, 256, 200704
g = torch.Generator(device=dev); g.manual_seed(3)
def make(kind):
    out = []
    for t in range(T):
        if kind == "normal":
            x = torch.randn(B, E, generator=g, device=dev)
        elif kind == "loguniform":
            x = torch.exp2(torch.rand(B, E, generator=g, device=dev) * 30.0 - 17.0)
        elif kind == "relu":
            x = torch.randn(B, E, generator=g, device=dev).clamp_(min=0)
        elif kind == "onebin":
            x = 1.0 + torch.rand(B, E, generator=g, device=dev) * 0.01
        elif kind == "octave":
            x = 1.0 + torch.rand(B, E, generator=g, device=dev)
        else:
            x = torch.zeros(B, E, device=dev)
        out.append(x)
    return out
plan = ops.TensorSetPlan([E] * T, B, dev)
for kind in (sys.argv[1:] or ("normal", "relu", "loguniform", "octave", "zeros")):
    data = make(kind)
    for form in ("oneread",):
        plan.octav_reset()
        for _ in range(3):
            ops.octav_batch(plan, data, False, form=form)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 6
        for _ in range(n):
            ops.octav_batch(plan, data, False, form=form)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gb = 4.0 * B * T * E / 1e9
        print(f"{kind:11s} {form}: {ms:.3f} ms/batch  {gb / ms * 1e3:.0f} GB/s", flush=True)
    del data
