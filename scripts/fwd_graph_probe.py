"""Is the network forward of the end-to-end path bound by the host issuing library calls?  Times ResNet-50 through
executor.GraphSession at several batch sizes: eager (host issue seconds and wall per batch) against one hipGraph replay
of the same forward (torch.cuda.CUDAGraph).  python scripts/fwd_graph_probe.py [model]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import models
from dipoorlet_amd.executor import GraphSession


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    g = getattr(models, name)()
    s = GraphSession(g)
    if "--untag" in sys.argv:     # the constants without their host copies: every Reshape / Gather / Clip reads the device back
        for t in s.consts.values():
            for a in ("_dpl_ints", "_dpl_floats"):
                if hasattr(t, a):
                    delattr(t, a)
    shape = [max(1, int(d)) for d in g.get_tensor_shape(s.input_names[0])]
    reps = 10
    for B in (16, 32, 64, 128):
        x = torch.randn([B * shape[0]] + shape[1:], device="cuda")
        feeds = {s.input_names[0]: x}
        for _ in range(3):
            out = s.run(feeds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = s.run(feeds)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eager = (t2 - t0) / reps
        issue = (t1 - t0) / reps
        del out
        print(f"{name} batch {B}: eager {eager * 1e3:.2f} ms/batch ({B / eager:.0f} img/s; host issue {issue * 1e3:.2f} ms)", flush=True)
        if "--no-graph" in sys.argv:
            continue
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = s.run(feeds)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        del out
        gr = torch.cuda.CUDAGraph()
        tc = time.perf_counter()
        with torch.cuda.graph(gr):
            out = s.run(feeds)
        torch.cuda.synchronize()
        tcap = time.perf_counter() - tc
        for _ in range(2):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / reps
        print(f"{name} batch {B}: "
              f"graph replay {rep * 1e3:.2f} ms/batch ({B / rep:.0f} img/s), capture {tcap * 1e3:.0f} ms", flush=True)
        del gr, out


if __name__ == "__main__":
    main()
