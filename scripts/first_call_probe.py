"""What do the FIRST calls of the library ops of a ResNet forward cost a fresh process (code-object loading, algorithm search)?
python scripts/first_call_probe.py"""
import time
import torch
import torch.nn.functional as F


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    y = fn()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} first {1e3 * (t1 - t0):8.1f} ms   second {1e3 * (t2 - t1):7.2f} ms", flush=True)
    return y


def main():
    t0 = time.perf_counter()
    x = torch.zeros(1, 3, 224, 224, device="cuda")
    torch.cuda.synchronize()
    print(f"context + first alloc          {1e3 * (time.perf_counter() - t0):8.1f} ms")
    w = torch.randn(64, 3, 7, 7, device="cuda")
    y = timed("conv2d 7x7 s2 (batch 1)", lambda: F.conv2d(x, w, None, 2, 3))
    timed("relu", lambda: F.relu(y))
    timed("max_pool2d 3x3 s2", lambda: F.max_pool2d(y, 3, 2, 1))
    timed("add", lambda: y + y)
    timed("adaptive_avg_pool / mean", lambda: y.mean((2, 3)))
    a, b = torch.randn(1, 2048, device="cuda"), torch.randn(2048, 1000, device="cuda")
    timed("addmm", lambda: torch.addmm(torch.zeros(1000, device="cuda"), a, b))
    x32 = torch.zeros(32, 3, 224, 224, device="cuda")
    y32 = timed("conv2d 7x7 s2 (batch 32)", lambda: F.conv2d(x32, w, None, 2, 3))
    timed("max_pool2d (batch 32)", lambda: F.max_pool2d(y32, 3, 2, 1))
    timed("abs().amax()", lambda: y32.abs().amax())
    timed("contiguous copy", lambda: y32.transpose(0, 1).contiguous())


if __name__ == "__main__":
    main()
