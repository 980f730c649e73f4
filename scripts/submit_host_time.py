"""Host time of OctavPipeline.submit per batch (does the host keep up with a 0.58 ms streaming kernel?).
DPL_OCTAV_LANES=1 DPL_SEED0=1234 python3 scripts/submit_host_time.py [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
dev = torch.device("cuda")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spec = resnet50_tensors()
elems, B = [e for _, e, _ in spec], 32
seed0 = int(os.environ.get("DPL_SEED0", "500"))
pool = [synth_activations(spec, B, dev, seed=seed0 + k) for k in range(17)]
plan = ops.TensorSetPlan(elems, B, dev)
pool = [plan.bind(p) for p in pool]
pipe = ops.OctavPipeline(False, dev)
for rep in range(3):
    plan.octav_reset()
    pipe.reset_stats()
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for b in range(nb):
        t = time.perf_counter()
        pipe.submit(plan, pool[b % 17])
        ts.append(time.perf_counter() - t)
    t_loop = time.perf_counter() - t0
    pipe.sync()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
ts_us = sorted(1e6 * x for x in ts)
print(f"submit host us: median {ts_us[len(ts_us) // 2]:.0f}, p90 {ts_us[int(0.9 * len(ts_us))]:.0f}, max {ts_us[-1]:.0f}; over 300 us: {[(i, round(1e6 * x)) for i, x in enumerate(ts) if x > 3e-4][:20]}")
print(f"host loop {1e3 * t_loop:.1f} ms, until the device is done {1e3 * t_all:.1f} ms = {1e3 * t_all / nb:.3f} ms/batch; compaction pairs {pipe.compaction_pairs}")
