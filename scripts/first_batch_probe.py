"""GPU: why does the FIRST forward of a warmed-up process take 87 ms on the GPU when the next ones take 9 ms?  Candidates: memory the
process has never touched (hipMalloc + the driver's clear), a GPU that has been idle (clocks), something else.  A fixed workload
(40 x relu over 205 MB into fresh outputs = what a forward's allocations look like) timed with events under each condition."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda")
torch.relu(torch.zeros(8, device=dev)); torch.cuda.synchronize()          # code object, context


def work(x, keep):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(40):
        keep.append(torch.relu(x))
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1), 2), round(host * 1e3, 2)


x = torch.zeros(64, 256, 56, 56, device=dev)
torch.cuda.synchronize()
keep = []
print("fresh memory (40 x 205 MB never allocated), GPU idle before:", work(x, keep))
keep.clear()
print("cached memory, GPU hot:", work(x, keep))
keep.clear()
time.sleep(0.5)
print("cached memory, GPU idle for 0.5 s:", work(x, keep))
keep.clear()
torch.cuda.empty_cache()
print("fresh memory again (cache emptied), GPU hot:", work(x, keep))
keep.clear()
torch.cuda.empty_cache()
time.sleep(0.5)
t0 = time.perf_counter()
big = torch.empty(41 * x.numel() * 4, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
print("one 8.4 GB allocation: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
del big
print("memory from ONE pre-allocated block (split by the caching allocator), GPU idle before:", work(x, keep))
