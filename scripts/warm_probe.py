"""Do the libraries' first calls overlap when made from several threads?  python scripts/warm_probe.py [serial|parallel]"""
import sys
import threading
import time

T0 = time.perf_counter()
import torch
import torch.nn.functional as F

mode = sys.argv[1] if len(sys.argv) > 1 else "parallel"
marks = {}


def mark(n):
    marks[n] = time.perf_counter() - t_start


def gemm():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        mark("gemm:begin")
        a = torch.zeros(8, 64, device="cuda")
        torch.addmm(torch.zeros(64, device="cuda"), a, torch.zeros(64, 64, device="cuda"))
        torch.cuda.current_stream().synchronize()
        mark("gemm:end")


def kernels():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        mark("kern:begin")
        x = torch.zeros(2, 8, 16, 16, device="cuda")
        torch.cuda.current_stream().synchronize()
        mark("kern:first")
        y = torch.relu(x); y = F.max_pool2d(y, 3, 2, 1); y = y + y; y.mean((2, 3), keepdim=True); y.abs().amax(); y.transpose(0, 1).contiguous()
        torch.cuda.current_stream().synchronize()
        mark("kern:elementwise")


def convs():
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        mark("conv:begin")
        for (cin, cout, k, s, hw) in [(3, 64, 7, 2, 224), (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 128, 1, 1, 56),
                                      (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28),
                                      (256, 1024, 1, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7)]:
            F.conv2d(torch.zeros(32, cin, hw, hw, device="cuda"), torch.zeros(cout, cin, k, k, device="cuda"), torch.zeros(cout, device="cuda"), s, k // 2)
        torch.cuda.current_stream().synchronize()
        mark("conv:end")


t_start = time.perf_counter()
torch.cuda.init()
torch.cuda.device_count()
mark("hipInit")
fns = [gemm, kernels, convs]
if mode == "serial":
    for f in fns:
        f()
else:
    ths = [threading.Thread(target=f) for f in fns]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
mark("all")
print(mode, "import %.3f" % (t_start - T0), {k: round(v, 3) for k, v in sorted(marks.items(), key=lambda kv: kv[1])})
