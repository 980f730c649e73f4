import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dipoorlet_amd import _hip, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
B = 4
sizes = [802816, 401408, 20480, 602112, 300000, 150528]
kinds = ["normal", "saturated", "dense_small", "const", "two_level", "relu"]
def draw(kind, n, scale):
    z = torch.randn(B, n, generator=g, device=dev) * scale
    if kind == "saturated": z = z.clamp_(-0.4 * scale, 0.4 * scale)
    elif kind == "dense_small": z = z.abs_() + 0.5
    elif kind == "const": z = torch.full((B, n), 1.25 * scale, device=dev)
    elif kind == "two_level": z = torch.where(z > 0, torch.full_like(z, 2.0 * scale), torch.full_like(z, 0.125))
    elif kind == "relu": z = z.clamp_(min=0)
    return z.contiguous()
batches = [[draw(k, n, 1.0 + 0.3 * it) for k, n in zip(kinds, sizes)] for it in range(6)]
plan = ops.TensorSetPlan(sizes, B, dev)
states = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
dt = np.dtype([("sum","<f8"),("cnt_gt","<u8"),("cnt_le","<u8"),("min","<u4"),("max","<u4"),("nan","<u4"),("done","<u4"),("s","<f4"),("ud","<f4"),("iters","<u4"),("mode","<u4"),("n","<u8"),("len0","<u4"),("len1","<u4"),("cur","<u4"),("res","<u4")])
for k, x in enumerate(batches):
    got = ops.octav_batch(plan, x, False, states, form="tail")
    st = np.frombuffer(states.cpu().numpy().tobytes(), dtype=dt)
    print(k, "ctl cnt_le", st["cnt_le"][-1], "rescued", st["len0"][-1], "modes", st["mode"][:-1].reshape(B, -1).tolist())
