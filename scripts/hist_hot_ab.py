"""GPU: k_abs_hist of two builds of the library A/B in ONE process under SUSTAINED load — the chip runs warm after a minute
and its shader clock comes down; a kernel at 67 % vector-unit busy then follows the clock where a plain streaming read does not.
python scripts/hist_hot_ab.py <a.so> <b.so> [seconds]  ->  per 10 s: median launch time of each build (alternating launches)"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import _hip, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations


def main():
    libs = []
    for path in sys.argv[1:3]:
        l = C.CDLL(path)
        name = "dpl_abs_hist_accumulate"
        fn = getattr(l, name)
        fn.restype, fn.argtypes = _hip.SIGNATURES[name]
        libs.append(fn)
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
    dev = torch.device("cuda:0")
    spec = resnet50_tensors()
    elems = [e for _, e, _ in spec]
    B = 32
    pool = [synth_activations(spec, B, dev, seed=7 + j) for j in range(4)]
    plan = ops.TensorSetPlan(elems, B, dev)
    acc = ops.CalibAccumulators(len(elems), dev, 2048)
    for p in pool:
        acc.minmax_accumulate(plan, p)
    acc.finalize_minmax()
    acc.hist_prepare()
    w = plan.work("hist")
    tabs = [plan.seg_table(p) for p in pool]
    torch.cuda.synchronize()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_end = time.time() + budget
    t0 = time.time()
    while time.time() < t_end:
        evs = [[], []]
        for i in range(400):        # ~0.2 s of back-to-back launches, the builds alternating
            k = i & 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            libs[k](*w.args(), ops._ptr(tabs[(i >> 1) % 4]), ops._ptr(acc.ranges), acc.bins, ops._ptr(acc.hist), stream)
            e1.record()
            evs[k].append((e0, e1))
        torch.cuda.synchronize()
        if int(time.time() - t0) % 10 == 0 or time.time() >= t_end:
            med = [sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2] * 1e3 for ev in evs]
            print(f"t = {time.time() - t0:5.0f} s   A {med[0]:6.1f} us   B {med[1]:6.1f} us   B / A {med[1] / med[0]:.3f}", flush=True)
            time.sleep(0.0)


if __name__ == "__main__":
    main()
