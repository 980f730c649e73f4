"""Fresh-process fixed costs on the GPU box: how to get 100 MB of initializers to the device, what the first calls of the
libraries cost and whether a helper thread can take them off the main thread.  python scripts/startup_probe.py <variant>"""
import sys
import threading
import time

import numpy as np
import torch

T0 = time.perf_counter()


def lap(name, t):
    print(f"{name:46s} {1e3 * (time.perf_counter() - t):8.1f} ms", flush=True)


variant = sys.argv[1] if len(sys.argv) > 1 else "upload"
t = time.perf_counter()
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
lap("context", t)
n = 25_600_000
rng = np.random.default_rng(0)
src = rng.standard_normal(n, dtype=np.float32)       # "the file's bytes"
parts = np.array_split(src, 161)

if variant == "upload":
    t = time.perf_counter(); h = torch.empty(n, dtype=torch.float32, pin_memory=True); lap("pinned alloc 102 MB", t)
    t = time.perf_counter()
    o = 0
    hv = h.numpy()
    for p in parts:
        hv[o:o + p.size] = p
        o += p.size
    lap("pack 161 arrays into it", t)
    t = time.perf_counter(); d = h.to("cuda", non_blocking=True); lap("async H2D issue", t)
    torch.cuda.synchronize(); lap("... done", t)
    t = time.perf_counter(); flat = np.concatenate(parts); lap("pack into a pageable buffer", t)
    t = time.perf_counter(); d2 = torch.from_numpy(flat).to("cuda"); torch.cuda.synchronize(); lap("pageable H2D 102 MB", t)
    t = time.perf_counter()
    ds = [torch.from_numpy(p).to("cuda") for p in parts]
    torch.cuda.synchronize(); lap("161 pageable H2D", t)
    t = time.perf_counter()
    rt = torch.cuda.cudart()
    rc = rt.cudaHostRegister(src.ctypes.data, src.nbytes, 0)
    lap(f"hostRegister 102 MB (rc {rc})", t)
    t = time.perf_counter()
    dflat = torch.empty(n, dtype=torch.float32, device="cuda")
    o = 0
    for p in parts:
        dflat[o:o + p.size].copy_(torch.from_numpy(p), non_blocking=True)
        o += p.size
    lap("161 async copies from registered memory: issue", t)
    torch.cuda.synchronize(); lap("... done", t)
    print("pinned?", torch.from_numpy(parts[3]).is_pinned(), "equal", bool((dflat.cpu() == torch.from_numpy(src)).all()))
    t = time.perf_counter(); h2 = torch.empty(n, dtype=torch.float32, pin_memory=True); lap("second pinned alloc 102 MB", t)

if variant in ("first", "thread"):
    import torch.nn.functional as F

    def warm():
        t = time.perf_counter()
        a = torch.zeros(8, 64, device="cuda")
        torch.addmm(torch.zeros(64, device="cuda"), a, torch.zeros(64, 64, device="cuda"))
        lap("  [warm] addmm", t)
        t = time.perf_counter()
        x = torch.zeros(2, 8, 16, 16, device="cuda")
        w = torch.zeros(8, 8, 3, 3, device="cuda")
        y = F.conv2d(x, w, None, 1, 1)
        y = F.max_pool2d(torch.relu(y), 3, 2, 1) ; y = y + y; y.mean((2, 3)); y.abs().amax(); y.transpose(0, 1).contiguous()
        torch.matmul(torch.zeros(2, 4, 8, 8, device="cuda"), torch.zeros(2, 4, 8, 8, device="cuda"))
        torch.softmax(y, -1); torch.erf(y); F.layer_norm(y, y.shape[-1:])
        torch.cuda.synchronize()
        lap("  [warm] conv / relu / pool / add / mean / ...", t)

    th = None
    if variant == "thread":
        th = threading.Thread(target=warm, daemon=True)
        th.start()
    t = time.perf_counter()
    h = torch.empty(n, dtype=torch.float32, pin_memory=True)
    hv = h.numpy()
    o = 0
    for p in parts:
        hv[o:o + p.size] = p
        o += p.size
    d = h.to("cuda", non_blocking=True)
    lap("main: pinned alloc + pack + H2D issue", t)
    if th is not None:
        t = time.perf_counter(); th.join(); lap("main: waited for the warm-up thread", t)
    sys.path.insert(0, ".")
    from dipoorlet_amd import models
    t = time.perf_counter(); g = models.resnet50(); lap("build graph", t)
    t = time.perf_counter(); s = g.make_session(); torch.cuda.synchronize(); lap("session", t)
    x = torch.randn(32, 3, 224, 224, device="cuda")
    for k in range(3):
        t = time.perf_counter(); outs = s._collect(s._run_env({"input": x}, 32), s.tensor_names, 32); torch.cuda.synchronize(); lap(f"forward B=32 #{k}", t)
lap("total since import", T0)
