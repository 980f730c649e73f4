"""ViT-B/16-like tensor sizes through the one-read form (both walks, pipeline) against the two-read bracket form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dipoorlet_amd import ops
dev = torch.device("cuda:0")
sizes = [150528] + [151296, 151296, 453888, 465708, 465708, 151296, 151296, 605184, 605184, 151296] * 6 + [768, 1000, 197, 3]
B = 8
g = torch.Generator(device=dev); g.manual_seed(5)
def batch(k):
    out = []
    for t, n in enumerate(sizes):
        x = torch.randn(B, n, generator=g, device=dev) * (0.5 + 0.05 * t) * (1.0 + 0.05 * torch.rand(B, 1, generator=g, device=dev))
        if t % 3 == 1:
            x = torch.nn.functional.gelu(x)
        if t % 5 == 4:
            x = torch.softmax(x.view(B, -1, 197)[:, : n // 197], -1).reshape(B, -1)[:, :n].contiguous() if n % 197 == 0 else x
        out.append(x.contiguous())
    return out
batches = [batch(k) for k in range(6)]
want = [ops.octav_batch(ops.TensorSetPlan(sizes, B, dev), x, False, form="bracket").cpu().numpy() for x in batches]
for walk in ("group", "sorted", "auto"):
    os.environ["DPL_OCTAV_WALK"] = walk
    plan = ops.TensorSetPlan(sizes, B, dev)
    pipe = ops.OctavPipeline(False, dev)
    outs = [pipe.submit(plan, x) for x in batches]
    pipe.sync()
    bad = 0
    for o, w in zip(outs, want):
        got = o.cpu().numpy()
        ok = np.array_equal(got[:, :, 1:], w[:, :, 1:], equal_nan=True) and np.allclose(got[:, :, 0], w[:, :, 0], rtol=2.4e-7, atol=0, equal_nan=True)
        bad += 0 if ok else 1
    print(walk, "batches differing from the bracket form:", bad, "of", len(outs), "| listed share %.3f, missed pairs %d, sorted batches %d, switched %s" % (pipe.list_share, pipe.fallback_pairs, pipe.sorted_batches, bool(pipe.switched)))
