"""GPU: the ViT-B/16 `-A mse` run (N = 256 in batches of 8, cold, two-stream pipeline) over images drawn with different seeds: how
many flat-distribution pairs (erf outputs: long lists) a batch holds moves its time by +- 8 %.  DPL_LIB=<other build>
for an A/B on one box.  python scripts/vit_seeds.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dipoorlet_amd import models, ops
dev = torch.device("cuda")
sess = models.vit_b16(seed=5, attn_gain=10.0).make_session()
elems, B = [int(e) for e in sess.elems_per_image], 8
plan = ops.TensorSetPlan(elems, B, dev)
pipe = ops.OctavPipeline(False, dev)
for seed in (1, 4242, 3):
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    pool = [[t.reshape(B, -1) for t in sess.run({"input": torch.randn(B, 3, 224, 224, generator=gen, device=dev)})] for _ in range(17)]
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.octav_reset()
        outs = [pipe.submit(plan, pool[b % 17]) for b in range(32)]
        pipe.sync()
        e1.record(); torch.cuda.synchronize()
    print(f"seed {seed}: {e0.elapsed_time(e1) / 32:.3f} ms/batch", flush=True)
    del pool
