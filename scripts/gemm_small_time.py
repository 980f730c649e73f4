"""GPU: dpl_gemm_small on ResNet-50's classifier head at the default batch beside hipBLASLt (torch.addmm), microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from dipoorlet_amd import ops
dev=torch.device("cuda")
a=torch.randn(64,2048,device=dev); w=torch.randn(1000,2048,device=dev); c=torch.randn(1000,device=dev)
for f,name in ((lambda: ops.gemm_small(a,w.t(),c),"gemm_small"),(lambda: torch.addmm(c,a,w.t()),"addmm")):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize(); print(name, e0.elapsed_time(e1)/100*1e3, "us")
