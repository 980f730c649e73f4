"""GPU: k_octav_tail of two builds of the library A/B in ONE process: the pipeline (streaming kernels one after the other on
the caller's stream, lanes = 1) runs its usual sequence through build A, except that every other batch's streaming kernel is
launched from build B (same job struct, same device state: the builds share the ABI).  Timed per launch by events on the stream.
python scripts/tail_hot_ab.py <b.so> [resnet50|vit] [jitter] [runs]   (build A = the library in the tree, or DPL_LIB)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import _hip, models, ops
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations


class Alternate:
    def __init__(self, a, b):
        self.a, self.b, self.n = a, b, 0

    def __getattr__(self, name):
        if name == "dpl_octav_oneread_stream":
            self.n += 1
            return getattr(self.b if self.n & 1 == 0 else self.a, name)
        return getattr(self.a, name)


def main():
    dev = torch.device("cuda")
    a = _hip.lib()
    b = C.CDLL(sys.argv[1])
    fn = b.dpl_octav_oneread_stream
    fn.restype, fn.argtypes = _hip.SIGNATURES["dpl_octav_oneread_stream"]
    _hip._lib = Alternate(a, b)
    which = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
    jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 7      # (1 warm-up + an even number)
    if which == "vit":
        sess = models.vit_b16(seed=5, attn_gain=10.0).make_session()
        elems, B = [int(e) for e in sess.elems_per_image], 8
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        pool = [[t.reshape(B, -1) for t in sess.run({"input": torch.randn(B, 3, 224, 224, generator=gen, device=dev)})] for _ in range(17)]
    else:
        spec = resnet50_tensors()
        elems, B = [e for _, e, _ in spec], 32
        pool = [synth_activations(spec, B, dev, seed=500 + k, image_jitter=jit) for k in range(17)]
    plan = ops.TensorSetPlan(elems, B, dev)
    pipe = ops.OctavPipeline(False, dev, lanes=1)
    pipe.record_events = True
    ta, tb = [], []
    for run in range(runs):
        plan.octav_reset()
        pipe.events = []
        _hip._lib.n = run & 1       # (which build takes the even batches swaps from run to run: the batches differ)
        outs = [pipe.submit(plan, pool[k % 17]) for k in range(64)]
        pipe.sync()
        torch.cuda.synchronize()
        ms = [e0.elapsed_time(e1) for e0, e1 in pipe.events]
        if run:     # (the first run warms up)
            ev, od = ms[16::2], ms[17::2]          # (the first batches of a cold run list more: left out on both sides)
            ta += od if run & 1 else ev            # call number n (1-based) goes to B when n + (run & 1) is even
            tb += ev if run & 1 else od
        del outs
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print(f"{which} jitter {jit}: A {med(ta):.1f} us  B {med(tb):.1f} us  B / A {med(tb) / med(ta):.4f}   ({len(ta)} + {len(tb)} launches)")


if __name__ == "__main__":
    main()
