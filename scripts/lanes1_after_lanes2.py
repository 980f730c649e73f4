"""Why is the one-stream OCTAV sweep slower when a two-lane pipeline ran before it in the same process (bench.py: mse, then mse_lanes1)?
python3 scripts/lanes1_after_lanes2.py <variant>   variants: l1 | l2_l1 | l2_del_l1 | l1state_l2_l1 | l2_l1_newplan"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
dev = torch.device("cuda")
variant = sys.argv[1]
spec = resnet50_tensors()
elems, B = [e for _, e, _ in spec], 32
raw = [synth_activations(spec, B, dev, seed=1234 + k) for k in range(17)]
plan = ops.TensorSetPlan(elems, B, dev)
pool = [plan.bind(p) for p in raw]


def sweep(pipe, plan, pool, tag, reps=3):
    pipe.record_events = True
    for rep in range(reps):
        plan.octav_reset()
        pipe.events.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = [pipe.submit(plan, pool[b % 17]) for b in range(128)]
        pipe.sync()
        e1.record()
        torch.cuda.synchronize()
    k = sorted(a.elapsed_time(b) for a, b in pipe.events)
    print(f"{variant:16s} {tag}: {e0.elapsed_time(e1) / 128:.4f} ms/batch; streaming kernel by events: median {k[len(k) // 2]:.4f} ms, mean {sum(k) / len(k):.4f}", flush=True)


if variant == "l1state_l2_l1":
    p1 = ops.OctavPipeline(False, dev, lanes=1)
    p1._state(plan, plan.octav_tail())
if variant != "l1":
    p2 = ops.OctavPipeline(False, dev, lanes=2)
    sweep(p2, plan, pool, "two lanes")
    if variant == "l2_del_l1":
        del p2
        gc.collect()
        torch.cuda.empty_cache()
if variant == "l2_l1_newplan":
    plan = ops.TensorSetPlan(elems, B, dev)
    pool = [plan.bind(p) for p in raw]
if variant != "l1state_l2_l1":
    p1 = ops.OctavPipeline(False, dev, lanes=1)
sweep(p1, plan, pool, "one stream")
