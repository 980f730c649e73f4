import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dipoorlet_amd import ops
dev = torch.device("cuda")
torch.zeros(1, device=dev); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1500000); e1.record(); torch.cuda.synchronize()
print("_sleep(1.5e6) takes %.3f ms" % e0.elapsed_time(e1))
main = torch.cuda.current_stream()
# which of 12 consecutive pool streams run beside the default stream, and beside each other?
ss = [torch.cuda.Stream(dev) for _ in range(12)]
print("ops._behind(pool stream, [default stream]):", [round(ops._behind(dev, s, [main]), 2) for s in ss])
print("ops._behind(pool stream, [pool stream 0]): ", [round(ops._behind(dev, s, [ss[0]]), 2) for s in ss])
t = time.perf_counter(); p = ops.OctavPipeline(False, dev, lanes=1); t1 = time.perf_counter() - t
t = time.perf_counter(); p2 = ops.OctavPipeline(False, dev, lanes=2); t2 = time.perf_counter() - t
print("pipeline creation: one stream %.1f ms, two lanes %.1f ms" % (1e3 * t1, 1e3 * t2))


def behind_reversed(stream, other, cycles=1500000):
    """the question the other way round — the spin on `other`, the marker on `stream` — which does not work against the default stream"""
    s0, s1, m = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(other):
        s0.record(other)
        torch.cuda._sleep(cycles)
        s1.record(other)
    m.record(stream)
    s1.synchronize()
    m.synchronize()
    return round(min(1.0, max(0.0, s0.elapsed_time(m) / max(s0.elapsed_time(s1), 1e-6))), 2)


print("roles swapped (spin on the default stream, marker on the pool stream):", [behind_reversed(s, main) for s in ss])
