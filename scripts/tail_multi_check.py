import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dipoorlet_amd import ops
from oracle import np_oracle as O
dev = torch.device("cuda")
rng = np.random.default_rng(5)
g = torch.Generator(device=dev); g.manual_seed(3)
def close(a, b): return (np.isnan(a) & np.isnan(b)) | (a == b) | (np.abs(a - b) <= 1e-5 * np.maximum(1.0, np.abs(b)))
B = 2
sizes = [1200007, 777, 3000000, 150528, 5000001, 2097152, 1044481]
kinds = ["normal", "relu", "uniform", "normal", "relu", "const", "heavy"]
def draw(kind, n, scale):
    z = torch.randn(B, n, generator=g, device=dev)
    if kind == "relu": z = z.clamp_(min=0)
    elif kind == "uniform": z = torch.rand(B, n, generator=g, device=dev) * 2 - 1
    elif kind == "const": z = torch.full((B, n), 0.37, device=dev)
    elif kind == "heavy": z = z * torch.exp(torch.randn(B, n, generator=g, device=dev))
    return (z * scale).contiguous()
bad = 0
for dyn in (False, True):
    plan = ops.TensorSetPlan(sizes, B, dev)
    pipe = ops.OctavPipeline(dyn, dev)
    batches = [[draw(k, n, 1.0 + 0.2 * it + 0.1 * t) for t, (k, n) in enumerate(zip(kinds, sizes))] for it in range(5)]
    outs = [pipe.submit(plan, x) for x in batches]
    pipe.sync(); torch.cuda.synchronize()
    print("dyn", dyn, "rescued", pipe.fallback_pairs, "compaction", pipe.compaction_pairs, "listed", round(pipe.list_share, 4))
    for it, (x, o) in enumerate(zip(batches, outs)):
        want = ops.octav_batch(ops.TensorSetPlan(sizes, B, dev), x, dyn, form="bracket").cpu().numpy()
        got = o.cpu().numpy()
        ok_mm = np.array_equal(got[..., 1:], want[..., 1:])
        ok_s = close(got[..., 0], want[..., 0]).all()
        if not (ok_mm and ok_s):
            bad += 1
            print("MISMATCH batch", it, "minmax", ok_mm, np.abs(got[..., 0] - want[..., 0]).max(), got[..., 0], want[..., 0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = batches[-1]; got = outs[-1].cpu().numpy()
        for t in (0, 2, 5):
            ref = O.octav_scale(x[t][1].cpu().numpy(), 4 if (dyn and abs(float(x[t][1].min())) < 1e-6) else 1)
            if not close(np.float32(got[1, t, 0]), np.float32(ref)):
                bad += 1; print("ORACLE MISMATCH", t, got[1, t, 0], ref)
print("bad", bad)
