"""GPU: the exact-tail OCTAV form against the numpy oracle and the two-read bracket form on a zoo of distributions and
sizes, cold (first call of a plan: thresholds raised on the fly) and warm (thresholds from the previous calls), with the
control block's statistics (listed share, rescued pairs, raises).  python scripts/tail_check.py"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy import special
from dipoorlet_amd import _hip, ops
from oracle import np_oracle as O

dev = torch.device("cuda")
rng = np.random.default_rng(5)


def gen(kind, n):
    sc = np.float32(10 ** rng.uniform(-1.5, 1.5))
    if kind == "normal": x = rng.standard_normal(n)
    elif kind == "relu": x = np.maximum(rng.standard_normal(n), 0)
    elif kind == "laplace": x = rng.laplace(0, 1, n)
    elif kind == "uniform": x = rng.uniform(-1, 1, n)
    elif kind == "student": x = rng.standard_t(3, n)
    elif kind == "lognormal": x = rng.lognormal(0, 1.0, n)
    elif kind == "erf": x = special.erf(rng.standard_normal(n))
    elif kind == "gelu":
        z = rng.standard_normal(n) * 2; x = 0.5 * z * (1 + special.erf(z / np.sqrt(2)))
    elif kind == "channels":
        C = 64; x = (rng.standard_normal((C, n // C + 1)) * rng.lognormal(0, 1.0, (C, 1))).ravel()[:n]
    elif kind == "hotfirst":
        C = 64; sc_c = np.ones((C, 1)); sc_c[:4] = 8.0; x = (rng.standard_normal((C, n // C + 1)) * sc_c).ravel()[:n]
    elif kind == "hotlast":
        C = 64; sc_c = np.ones((C, 1)); sc_c[-4:] = 8.0; x = (rng.standard_normal((C, n // C + 1)) * sc_c).ravel()[:n]
    elif kind == "relu6": x = np.clip(rng.standard_normal(n) * 3, 0, 6)
    elif kind == "const": x = np.where(rng.random(n) < 0.5, 2.0, 0.0)
    elif kind == "tanh": x = np.tanh(rng.standard_normal(n) * 3)
    else: raise ValueError(kind)
    return (np.asarray(x, np.float64) * sc).astype(np.float32)


KINDS = ["normal", "relu", "laplace", "uniform", "student", "lognormal", "erf", "gelu", "channels", "hotfirst", "hotlast", "relu6", "const", "tanh"]
sizes = [802816, 401408, 200704, 100352, 50176, 25088, 20481, 20480, 2048, 1000, 605184, 150528, 1044480, 77777]
B = 3
bad = 0
for rep, n in enumerate(sizes):
    kinds = [KINDS[(rep + i) % len(KINDS)] for i in range(5)]
    raw = [np.stack([gen(k, n) for _ in range(B)]) for k in kinds]
    # images of a tensor differ in scale from call to call (what the threshold history has to live with)
    plan = ops.TensorSetPlan([n] * len(kinds), B, dev)
    states = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
    for call, scale in enumerate([1.0, 1.0, 0.8, 1.3, 1.0]):
        data = [(r * np.float32(scale)).astype(np.float32) for r in raw]
        tensors = [torch.from_numpy(d).to(dev) for d in data]
        got = ops.octav_batch(plan, tensors, False, states, form="tail").cpu().numpy()
        ctl = _hip.OctavState.from_buffer_copy(states.cpu().numpy()[-80:].tobytes())
        ref = ops.octav_batch(ops.TensorSetPlan([n] * len(kinds), B, dev), tensors, False, form="bracket").cpu().numpy()
        worst = 0.0
        for t, k in enumerate(kinds):
            for b in range(B):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    want = float(O.octav_scale(data[t][b], 1))
                g = float(got[b, t, 0])
                err = abs(g - want) / max(1.0, abs(want))
                worst = max(worst, err)
                if not (err <= 1e-5) or got[b, t, 1] != data[t][b].min() or got[b, t, 2] != data[t][b].max():
                    bad += 1
                    print("  MISMATCH", n, k, b, call, g, want, float(ref[b, t, 0]))
        same = int((got[:, :, 0] == ref[:, :, 0]).sum())
        print(f"n={n} call={call} scale={scale}: listed {ctl.sum / (n * B * len(kinds)):.4f} rescued {ctl.len0} compaction {ctl.cnt_le} "
              f"raises {ctl.iters} worst {worst:.2e} bit-equal to bracket {same}/{got[:, :, 0].size}")
print("BAD", bad)
sys.exit(1 if bad else 0)
