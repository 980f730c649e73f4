#!/usr/bin/env python3
"""Kernel micro-benchmark for tuning (GPU box).  One process per variant (the knobs are read once at
library load), several interleaved rounds; prints median / min kernel time and achieved GB/s.

  python scripts/kbench.py --kernel hist --variants 0,1,2,3 --rounds 3
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import torch

    from dipoorlet_amd import ops
    from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
    dev = torch.device("cuda:0")
    spec = resnet50_tensors()
    elems = [e for _, e, _ in spec]
    pool = [synth_activations(spec, a.batch, dev, seed=7 + j) for j in range(2)]
    plan = ops.TensorSetPlan(elems, a.batch, dev, chunk_elems=a.chunk or None)
    acc = ops.CalibAccumulators(len(elems), dev, a.bins)
    for p in pool:
        acc.minmax_accumulate(plan, p)
    acc.finalize_minmax()
    acc.hist_prepare()
    states = None
    fn = {"hist": lambda i: acc.abs_hist_accumulate(plan, pool[i % 2]),
          "minmax": lambda i: acc.minmax_accumulate(plan, pool[i % 2]),
          "octav": lambda i: ops.octav_batch(plan, pool[i % 2], False, states)}[a.kernel]
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for i in range(a.iters):  # back to back, no host sync in between (an idle GPU clocks down)
        evs[i][0].record()
        fn(i)
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs[a.iters // 4:])
    nbytes = 4 * sum(elems) * a.batch
    print(json.dumps({"median_ms": ts[len(ts) // 2], "min_ms": ts[0], "gbps_median": nbytes / ts[len(ts) // 2] / 1e6,
                      "items": plan.work({"octav": "octav"}.get(a.kernel, a.kernel), a.kernel == "octav").n_blocks,
                      "chunk": plan.chunk}))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--kernel", default="hist")
    p.add_argument("--variants", default="0")
    p.add_argument("--env", default="DPL_HIST_VARIANT")
    p.add_argument("--rounds", type=int, default=3)
    p.add_argument("--iters", type=int, default=80)
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--bins", type=int, default=2048)
    p.add_argument("--chunk", type=int, default=0)
    p.add_argument("--child", action="store_true")
    a = p.parse_args()
    if a.child:
        return child(a)
    res = {}
    for r in range(a.rounds):
        for v in a.variants.split(","):
            env = dict(os.environ)
            env[a.env] = v
            out = subprocess.run([sys.executable, __file__, "--child", "--kernel", a.kernel, "--iters", str(a.iters),
                                  "--batch", str(a.batch), "--bins", str(a.bins), "--chunk", str(a.chunk)],
                                 env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print("variant", v, "failed:", out.stderr[-500:])
                continue
            res.setdefault(v, []).append(d)
            print(f"round {r} {a.env}={v}: median {d['median_ms']*1e3:.1f} us  min {d['min_ms']*1e3:.1f} us  "
                  f"{d['gbps_median']:.0f} GB/s  items {d['items']} chunk {d['chunk']}", flush=True)
    for v, ds in res.items():
        m = sorted(x["median_ms"] for x in ds)
        print(f"== {a.env}={v}: median-of-medians {m[len(m)//2]*1e3:.1f} us, best min {min(x['min_ms'] for x in ds)*1e3:.1f} us")


if __name__ == "__main__":
    main()
