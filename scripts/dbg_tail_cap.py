import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dipoorlet_amd import _hip, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(23)
B, sizes = 3, [150528, 40000, 802816, 1000]
tensors = [torch.from_numpy(np.stack([(rng.standard_normal(n) * (1 + t)).astype(np.float32) if t % 2 == 0 else
                                      np.maximum(rng.standard_normal(n), 0).astype(np.float32) * 2.5
                                      for _ in range(B)])).to(dev) for t, n in enumerate(sizes)]
plan = ops.TensorSetPlan(sizes, B, dev)
for _ in range(3):
    want = ops.octav_batch(plan, tensors, False, form="tail").cpu().numpy()
states = torch.empty((plan.n_pairs + 1) * 80, dtype=torch.uint8, device=dev)
for rf in (0, 2):
    old = _hip.lib().dpl_test_hook_exact_fail_every(2); old_r = _hip.lib().dpl_test_hook_rescue_fail_every(rf)
    got = ops.octav_batch(plan, tensors, False, states, form="tail").cpu().numpy()
    ctl = _hip.OctavState.from_buffer_copy(states.cpu().numpy()[-80:].tobytes())
    _hip.lib().dpl_test_hook_exact_fail_every(old); _hip.lib().dpl_test_hook_rescue_fail_every(old_r)
    print("rescue_fail", rf, "len0", ctl.len0, "len1", ctl.len1, "cnt_le", ctl.cnt_le, "listed", ctl.sum, "maxdiff", np.nanmax(np.abs(got[..., 0] - want[..., 0])))
    st = np.frombuffer(states.cpu().numpy().tobytes(), dtype=np.dtype([("sum","<f8"),("cnt_gt","<u8"),("cnt_le","<u8"),("min","<u4"),("max","<u4"),("nan","<u4"),("done","<u4"),("s","<f4"),("ud","<f4"),("iters","<u4"),("mode","<u4"),("n","<u8"),("len0","<u4"),("len1","<u4"),("cur","<u4"),("res","<u4")]))
    print("  modes", st["mode"][:-1].tolist(), "done", st["done"][:-1].tolist(), "len0", st["len0"][:-1].tolist())
