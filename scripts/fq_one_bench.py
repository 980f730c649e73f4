import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import ops
dev = torch.device("cuda")
for shape in [(32, 256, 56, 56), (32, 64, 112, 112), (32, 2048, 7, 7), (32, 512, 28, 28)]:
    x = torch.randn(shape, device=dev); y = torch.empty_like(x)
    c = shape[1]
    s1, z1 = torch.full((1,), 0.05, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    sc, zc = torch.full((c,), 0.05, device=dev), torch.zeros(c, dtype=torch.int32, device=dev)
    for mode in ("tensor", "channel"):
        f = (lambda: ops.fake_quant(x, s1, z1, -128, 127, out=y)) if mode == "tensor" else (lambda: ops.fake_quant(x, sc, zc, -128, 127, axis=1, out=y))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(shape, mode, f"{ms*1e3:.1f} us  {8 * x.numel() / ms / 1e6:.0f} GB/s = {8 * x.numel() / ms / 1e6 / 8000:.3f}")
