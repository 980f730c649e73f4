"""GPU: does the FIRST device allocation of a process (the HIP context: 0.1 s) stall other Python threads — and is it the GIL?
A ticker thread records its longest pause while another thread makes the first allocation (a) through torch.zeros, (b) through a
ctypes call into the HIP runtime (ctypes releases the GIL) followed by torch.zeros.  python scripts/context_gil_probe.py [torch|ctypes] [pure|alloc]"""
import os, sys, threading, time
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
import torch
torch.cuda.set_device(0)          # hipInit, no context yet
import numpy as np
mode2 = sys.argv[2] if len(sys.argv) > 2 else "pure"
junk = []
ticks = []
stop = False


def first_alloc():
    t0 = time.perf_counter()
    if mode == "ctypes":
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        p = ctypes.c_void_p()
        hip.hipSetDevice(0)
        hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20))
        hip.hipDeviceSynchronize()
        hip.hipFree(p)
    t1 = time.perf_counter()
    x = torch.zeros(2, 8, 16, 16, device="cuda")
    torch.cuda.synchronize()
    print("%s: runtime call %.1f ms, torch.zeros + sync %.1f ms" % (mode, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))


th = threading.Thread(target=first_alloc)
t_start = time.perf_counter()
th.start()
while th.is_alive():
    ticks.append(time.perf_counter())
    if mode2 == "alloc":       # work that takes fresh memory from the C allocator (page faults, heap growth)
        junk.append(torch.from_numpy(np.empty(4096, np.float32)).clone())
        if len(junk) > 2000:
            junk.clear()
    else:
        sum(range(200))            # pure Python work
gaps = [b - a for a, b in zip(ticks, ticks[1:])]
print("main thread (" + mode2 + ") meanwhile: %d ticks, longest pause %.1f ms, pauses > 2 ms: %s"
      % (len(ticks), max(gaps) * 1e3, [round(g * 1e3, 1) for g in gaps if g > 2e-3]))
