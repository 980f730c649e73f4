import sys, os, numpy as np, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dipoorlet_amd import ops, _hip
from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
dev = torch.device("cuda:0")
spec = resnet50_tensors(); elems = [e for _, e, _ in spec]; T = len(elems)
B = 32
pool = [synth_activations(spec, B, dev, seed=1234 + j) for j in range(4)]
plan = ops.TensorSetPlan(elems, B, dev)
dt = np.dtype([("sum","<f8"),("cnt_gt","<u8"),("cnt_le","<u8"),("min_enc","<u4"),("max_enc","<u4"),("nan","<u4"),("done","<u4"),("s","<f4"),("ud","<f4"),("iters","<u4"),("mode","<u4"),("n","<u8"),("len0","<u4"),("len1","<u4"),("cur","<u4"),("res","<u4")])
import ctypes
L = _hip.lib()
for k in range(20):
    st = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
    # run only the one-read kernels' part by calling the op and inspecting state AFTER (fallback already finished: mode 1 pairs keep mode 1)
    ops.octav_batch(plan, pool[k % 4], False, st, form="oneread")
    torch.cuda.synchronize()
    a = st.cpu().numpy().view(dt)
    pairs = a[:-1]; ctl = a[-1]
    fb = np.nonzero(pairs["mode"] == 1)[0]
    sizes = collections.Counter(elems[p % T] for p in fb)
    tens = collections.Counter(int(p % T) for p in fb)
    print(f"batch {k}: fallback pairs {len(fb)} (ctl {ctl['cnt_le']}), by size {dict(sizes)}; tensors most hit {tens.most_common(5)}")
res = plan.octav_oneread_scratch()
vis = res["vis"].cpu().numpy().astype(np.uint32)
pc = np.array([[bin(int(w)).count("1") for w in (vis[0][t] | vis[1][t])] for t in range(T)]).sum(1)
print("marked bins per tensor (last written buffer):", pc.min(), np.median(pc), pc.max())
