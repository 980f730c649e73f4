import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations, resnet50_tensor_shapes
dev = torch.device("cuda")
spec = resnet50_tensors(); elems = [e for _, e, _ in spec]; B = 32
xs = synth_activations(spec, B, dev, seed=1)
ys = [torch.empty_like(x) for x in xs]
shapes = resnet50_tensor_shapes()
for nb in ([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (512, 1024, 2048, 4096, 8192)):
    os.environ["DPL_BLOCKS_FQ"] = str(nb)
    plan = ops.TensorSetPlan(elems, B, dev)
    for mode in ("t", "c"):
        prm = [((torch.full((1,), .05), torch.zeros(1, dtype=torch.int32), 1, -128, 127) if mode == "t" else
                (torch.full((c,), .05), torch.zeros(c, dtype=torch.int32), h * w, -128, 127)) for (c, h, w) in shapes]
        f = ops.FakeQuantSet(plan, prm)
        for _ in range(2): f(xs, out=ys)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f(xs, out=ys)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(nb, mode, f"{ms:.3f} ms  {8 * sum(elems) * B / ms / 1e6 / 8000:.3f}")

big = torch.randn(851_000_000, device=dev); out = torch.empty_like(big)
for _ in range(2): out.copy_(big)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): out.copy_(big)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"torch copy of 3.4 GB: {ms:.3f} ms  {8 * big.numel() / ms / 1e6 / 8000:.3f} of 8 TB/s (read + write)")
