"""Is the first-use cost of a convolution configuration paid per process or per thread (MIOpen handle)?"""
import threading
import time

import torch
import torch.nn.functional as F

CFG = [(3, 64, 7, 2, 224), (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (256, 128, 1, 1, 56),
       (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (512, 128, 1, 1, 28), (128, 128, 3, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28),
       (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14),
       (512, 2048, 1, 1, 7), (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7), (256, 512, 1, 2, 56), (512, 1024, 1, 2, 28), (1024, 2048, 1, 2, 14)]


def convs(tag, cfgs, B=32):
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        t = time.perf_counter()
        for (cin, cout, k, s, hw) in cfgs:
            F.conv2d(torch.zeros(B, cin, hw, hw, device="cuda"), torch.zeros(cout, cin, k, k, device="cuda"), torch.zeros(cout, device="cuda"), s, k // 2)
        torch.cuda.current_stream().synchronize()
        print(f"{tag:40s} {1e3 * (time.perf_counter() - t):7.1f} ms", flush=True)


torch.zeros(1, device="cuda")
torch.cuda.synchronize()
convs("main: one small conv (library init)", [(8, 8, 3, 1, 16)], 2)
th = threading.Thread(target=convs, args=("thread A: all configs, first time", CFG))
th.start(); th.join()
convs("main: all configs after thread A", CFG)
convs("main: again", CFG)
th = threading.Thread(target=convs, args=("thread B: all configs", CFG))
th.start(); th.join()
# fresh configs (batch 64), split over 3 threads
t = time.perf_counter()
ths = [threading.Thread(target=convs, args=(f"thread {i}: a third of the B=64 configs", CFG[i::3], 64)) for i in range(3)]
[x.start() for x in ths]; [x.join() for x in ths]
print("3 threads wall %.1f ms" % (1e3 * (time.perf_counter() - t)))
convs("main: B=64 configs after the 3 threads", CFG, 64)
convs("main: B=16 configs, cold, one thread", CFG, 16)
