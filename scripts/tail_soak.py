"""GPU soak of the exact-tail OCTAV form: random tensor sets (sizes 1 .. 3 500 000 incl. odd ones and pairs above one slice, 16 distribution kinds incl.
saturating / discrete / constant / heavy-tailed / per-channel-scaled, per-image scale jitter up to x 4, both dynamic_sym
settings) through ops.OctavPipeline over runs of batches (threshold history, raises on the fly, rescues, the compaction route) —
every pair against the two-read form on the GPU (which walks the reference's whole iterate sequence), a sample of pairs against
the numpy oracle.  python scripts/tail_soak.py [seconds] [seed]   ->   one summary line; exit 1 on any mismatch."""
import os
import sys
import time
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dipoorlet_amd import ops
from oracle import np_oracle as O


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)

    def draw(kind, b, n):
        z = torch.randn(b, n, generator=g, device=dev)
        if kind == 0: x = z
        elif kind == 1: x = z.clamp_(min=0)
        elif kind == 2: x = torch.rand(b, n, generator=g, device=dev) * 2 - 1
        elif kind == 3: x = torch.erf(z)
        elif kind == 4: x = z * torch.exp(torch.randn(b, n, generator=g, device=dev))          # heavy tails
        elif kind == 5: x = (z * 3).clamp_(0, 6)                                                 # saturating
        elif kind == 6: x = torch.round(z * 2) * 0.25                                            # discrete
        elif kind == 7: x = torch.full((b, n), float(rng.uniform(0.1, 3)), device=dev)           # constant
        elif kind == 8: x = torch.where(torch.rand(b, n, generator=g, device=dev) < 0.02, z * 5, torch.zeros_like(z))   # sparse
        elif kind == 9: x = torch.tanh(z * 3)
        elif kind == 10: x = torch.sigmoid(z * 4)
        elif kind == 11: x = z.abs() + 1e-7
        elif kind == 12:                                                                          # channel-major, per-channel scales
            c = 16
            sc = torch.exp(torch.randn(c, 1, generator=g, device=dev))
            x = (z[:, : (n // c) * c].reshape(b, c, -1) * sc).reshape(b, -1)
            x = torch.cat([x, z[:, x.shape[1]:]], 1)
        elif kind == 13: x = z * 1e-3
        elif kind == 14: x = z * 300.0
        else: x = 0.5 * z * (1 + torch.erf(z / 2 ** 0.5))                                        # GELU
        return x.contiguous()
    close = lambda a, b: (np.isnan(a) & np.isnan(b)) | (a == b) | (np.abs(a - b) <= 1e-5 * np.maximum(1.0, np.abs(b)))
    t_end = time.time() + budget
    pairs = bad = checked_np = 0
    worst = 0.0
    rescued = compaction = 0
    sets = 0
    while time.time() < t_end:
        T = int(rng.integers(3, 9))
        B = int(rng.integers(1, 5))
        sizes = [int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 60000), rng.integers(60000, 1044481), rng.integers(60000, 1044481),
                                 rng.integers(1044481, 3500000)])) for _ in range(T)]     # (one in five: a pair above one slice)
        kinds = [int(rng.integers(0, 16)) for _ in range(T)]
        dyn = bool(rng.random() < 0.3)
        plan = ops.TensorSetPlan(sizes, B, dev)
        ref_plan = ops.TensorSetPlan(sizes, B, dev)
        pipe = ops.OctavPipeline(dyn, dev)
        base = [draw(k, B, n) for k, n in zip(kinds, sizes)]
        batches, rows = [], []
        for k in range(int(rng.integers(2, 6))):
            jit = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
            f = torch.exp2((torch.rand(B, 1, generator=g, device=dev) * 2 - 1) * 2 * jit)        # per-image scale, up to x 4
            x = [(t * f).contiguous() if kd not in (3, 9, 10) else t for t, kd in zip(base, kinds)]
            batches.append(x)
            rows.append(pipe.submit(plan, x))
        pipe.sync()
        torch.cuda.synchronize()
        rescued += pipe.fallback_pairs
        compaction += pipe.compaction_pairs
        for x, r in zip(batches, rows):
            got = r.cpu().numpy()
            ref = ops.octav_batch(ref_plan, x, dyn, form="bracket").cpu().numpy()
            ok = close(got[..., 0].astype(np.float64), ref[..., 0].astype(np.float64)) & (got[..., 1] == ref[..., 1]) & (got[..., 2] == ref[..., 2])
            d = np.abs(got[..., 0].astype(np.float64) - ref[..., 0]) / np.maximum(1.0, np.abs(ref[..., 0]))
            worst = max(worst, float(np.nanmax(np.where(np.isfinite(d), d, 0.0))))
            pairs += ok.size
            if not ok.all():
                bad += int((~ok).sum())
                idx = np.argwhere(~ok)[0]
                print("MISMATCH", sizes, kinds, dyn, idx, got[tuple(idx)], ref[tuple(idx)], flush=True)
            t = int(rng.integers(0, T))
            b = int(rng.integers(0, B))
            xs = x[t][b].cpu().numpy()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = float(O.octav_scale(xs, O.octav_unsigned(xs.min(), dyn)))
            checked_np += 1
            if not close(np.float64(got[b, t, 0]), np.float64(want)):
                bad += 1
                print("ORACLE MISMATCH", sizes[t], kinds[t], dyn, got[b, t], want, flush=True)
        sets += 1
    print(f"tail_soak seed {seed}: {sets} tensor sets, {pairs} pairs against the two-read form + {checked_np} against the numpy oracle, "
          f"{bad} mismatches, worst relative difference {worst:.2e}; rescued {rescued} pairs, compaction route {compaction}")
    return 1 if bad else 0


if __name__ == "__main__":
    rc = main()
    torch.cuda.synchronize()
    sys.exit(rc)
