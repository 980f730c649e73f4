"""GPU: ResNet-50 forward through the executor at batch 64 — MIOpen's immediate mode (the default) against its search
(torch.backends.cudnn.benchmark = True: every convolution configuration is benchmarked once): steady milliseconds per batch and what
the first forward costs.  python scripts/conv_search_probe.py [0|1] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import models
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == "1"
g = models.resnet50()
s = g.make_session()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = {"input": torch.randn(B, 3, 224, 224, device="cuda")}
torch.cuda.synchronize()
t0 = time.perf_counter()
s.run(x)
torch.cuda.synchronize()
first = time.perf_counter() - t0
for _ in range(3):
    s.run(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    s.run(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("search" if torch.backends.cudnn.benchmark else "immediate", "first forward %.2f s, steady %.2f ms per %d images = %.0f images/s" % (first, ms, B, B * 1e3 / ms))
