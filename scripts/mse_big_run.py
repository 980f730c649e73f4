"""GPU: a cold `-A mse` run over a tensor set with pairs ABOVE one slice (the ResNet-50 set at 448 x 448 input: every tensor four
times its 224 x 224 size, up to 3.2 M elements per image; batches of 8) through ops.OctavPipeline.
python scripts/mse_big_run.py [batches] [jitter]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations


def main():
    dev = torch.device("cuda")
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    spec = [(n, 4 * e, k) for n, e, k in resnet50_tensors()]
    elems, B = [e for _, e, _ in spec], 8
    pool = [synth_activations(spec, B, dev, seed=500 + k, image_jitter=jit) for k in range(17)]
    plan = ops.TensorSetPlan(elems, B, dev)
    n_multi = plan.octav_tail().sizes.n_multi
    pipe = ops.OctavPipeline(False, dev)
    for rep in range(2):
        plan.octav_reset()
        pipe.reset_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = [pipe.submit(plan, pool[b % 17]) for b in range(nb)]
        pipe.sync()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / nb
    gb = 4 * sum(elems) * B / 1e9
    print(f"resnet50 @ 448 (pairs above one slice: {n_multi} of {plan.n_pairs}), jitter {jit}: {ms:.3f} ms/batch, {gb / ms * 1e3:.0f} GB/s credited = "
          f"{gb / ms / 8:.3f} of 8 TB/s; rescued/batch {pipe.fallback_pairs / pipe.batches:.1f} compaction {pipe.compaction_pairs} listed {pipe.list_share:.4f}")


if __name__ == "__main__":
    main()
