import sys, os, numpy as np, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dipoorlet_amd import ops, _hip
from oracle import np_oracle as O
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1000]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
xs = [torch.from_numpy(rng.standard_normal((B, n)).astype(np.float32)).to(dev) for n in sizes]
plan = ops.TensorSetPlan(sizes, B, dev)
print("slices", plan.octav_oneread_scratch()["n_slices"], flush=True)
for form in ("bracket", "oneread", "oneread", "oneread"):
    out = ops.octav_batch(plan, xs, False, form=form)
    torch.cuda.synchronize()
    print(form, out.cpu().numpy().reshape(-1, 3)[:6], flush=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    print("oracle", [O.octav_scale(x[0].cpu().numpy(), 1) for x in xs])
