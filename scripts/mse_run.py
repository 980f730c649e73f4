"""GPU: a cold `-A mse` run of N batches through ops.OctavPipeline on one of the two BASELINE tensor sets, timed by HIP events.
python scripts/mse_run.py [resnet50|vit] [batches] [pool] — environment (DPL_*) passes through; meant to sit under rocprofv3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import models, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations



def main():
    """(Everything lives in this function: module-level tensors, streams and pinned buffers destroyed at interpreter shutdown —
    after the profiler's tool library has gone — made `rocprofv3 --pmc` runs of this script end in a segmentation fault.)"""
    dev = torch.device("cuda")
    which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    npool = int(sys.argv[3]) if len(sys.argv) > 3 else 17
    jit = float(os.environ.get("DPL_BENCH_JITTER", "0"))
    if which == "vit":
        sess = models.vit_b16(seed=5, attn_gain=10.0).make_session()
        elems, B = [int(e) for e in sess.elems_per_image], 8
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        pool = [[t.reshape(B, -1) for t in sess.run({"input": torch.randn(B, 3, 224, 224, generator=gen, device=dev)})] for _ in range(npool)]
    elif which == "resnet50_real":     # ResNet-50 run by the repo's executor (random weights, random images): real layer statistics
        sess = models.resnet50().make_session()
        elems, B = [int(e) for e in sess.elems_per_image], 32
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        scale = lambda k: 1.0 + jit * (2.0 * torch.rand(B, 1, 1, 1, generator=gen, device=dev) - 1.0)
        pool = [[t.reshape(B, -1) for t in sess.run({"input": torch.randn(B, 3, 224, 224, generator=gen, device=dev) * scale(k)})]
                for k in range(npool)]
    else:
        spec = resnet50_tensors()
        elems, B = [e for _, e, _ in spec], 32
        pool = [synth_activations(spec, B, dev, seed=int(os.environ.get("DPL_SEED0", "500")) + k, image_jitter=jit) for k in range(npool)]
    plan = ops.TensorSetPlan(elems, B, dev)
    pool = [plan.bind(p) for p in pool]       # resident sets: validated and pinned once
    pipe = ops.OctavPipeline(False, dev)
    for rep in range(2):
        plan.octav_reset()
        pipe.reset_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if os.environ.get("DPL_SINGLE"):     # one stream, kernels back to back: their durations ALONE
            outs = [ops.octav_batch(plan, pool[b % npool], False) for b in range(nb)]
        else:
            outs = [pipe.submit(plan, pool[b % npool]) for b in range(nb)]
            pipe.sync()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / nb
    gb = 4 * sum(elems) * B / 1e9
    if os.environ.get("DPL_SINGLE"):
        print(f"{which} (single stream): {ms:.3f} ms/batch")
        return
    print(f"{which}: {ms:.3f} ms/batch, {gb / ms * 1e3:.0f} GB/s credited = {gb / ms / 8:.3f} of 8 TB/s; misses/batch {pipe.fallback_pairs / pipe.batches:.1f} "
          f"compaction {pipe.compaction_pairs} listed {pipe.list_share:.4f} "
          f"raises/batch {pipe.raises / pipe.batches:.1f} tiles twice/batch {pipe.tiles_reread / pipe.batches:.0f}")


if __name__ == "__main__":
    main()
    torch.cuda.synchronize()
