"""GPU: k_fake_quant's chunk per workgroup in elements (DPL_FQ_CHUNK) on the tensors a fake-quantised ResNet-50 forward at batch 64 runs it on
— launches over DISTINCT buffers in rotation (more than the 256 MB Infinity Cache between two uses of one).
python scripts/fq_blocks_ab.py   (spawns one child per setting)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from dipoorlet_amd import ops
    dev = torch.device("cuda")
    out = []
    for shape in [(64, 256, 56, 56), (64, 512, 28, 28), (64, 1024, 14, 14), (64, 2048, 7, 7), (64, 64, 56, 56), (64, 128, 28, 28)]:
        n = 1
        for d in shape:
            n *= d
        k = max(2, int(1.2e9 // (4 * n)))          # >= 1.2 GB of distinct inputs
        xs = [torch.randn(shape, device=dev) for _ in range(k)]
        y = torch.empty_like(xs[0])
        s1, z1 = torch.full((1,), 0.05, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        for x in xs:
            ops.fake_quant(x, s1, z1, -128, 127, out=y)
        reps = 4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for x in xs:
                ops.fake_quant(x, s1, z1, -128, 127, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * k)
        out.append(f"{8 * n / us / 1e3 / 8000:.3f}")
        del xs, y
    print(os.environ.get("DPL_FQ_CHUNK", "3072").rjust(6), " ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        print("chunk   [64,256,56,56] [64,512,28,28] [64,1024,14,14] [64,2048,7,7] [64,64,56,56] [64,128,28,28]  (fraction of 8 TB/s, read + write)")
        for rep in range(2):
            for b in ("1024", "2048", "3072", "4096", "8192", "12288"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, DPL_FQ_CHUNK=b))
