"""GPU: the AdaRound / BRECQ / QDrop kernels (dipoorlet_amd/csrc/round_kernels.hip through the C ABI) against the
reference-generated vectors (tests/golden/round_level.*) and against oracle/round_oracle.py on further seeds.
Tolerances: fp32 elementwise results 2e-6 relative (device expf / logf / powf are within a few ulp of the host's);
trajectories as in tests/test_round_oracle_golden.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "round_level.npz"))
META = json.load(open(os.path.join(HERE, "golden", "round_level.json")))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(a, b, rtol=2e-6, atol=1e-7):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def mask_gradient(rp, G, reg_beta=0.0):
    """dL/d(mask) for upstream dL/d(qw) = G through the fused step kernel in gradients-only mode."""
    from dipoorlet_amd import _hip
    from dipoorlet_amd.ops import _ptr, _stream
    from dipoorlet_amd.weight_transform.ada_quant_layer import _step_params
    g = torch.empty_like(rp.round_mask)
    val = torch.zeros(1, dtype=torch.float64, device="cuda")
    p = _step_params(adam=0, clamp=rp.clamp, reg_beta=reg_beta)
    _hip.check(_hip.lib().dpl_round_step(_ptr(G) if G is not None else None, _ptr(rp.wfloor), _ptr(rp.round_mask), None,
                                         None, _ptr(rp.scale), _ptr(rp.q_min), _ptr(rp.q_max), rp.n, rp.nch, rp.inner,
                                         C.byref(p), None, None, _ptr(g), _ptr(val), _stream()), "dpl_round_step")
    return g, float(val[0])


@pytest.mark.parametrize("case", META["quant_weight"], ids=lambda c: c["key"])
def test_quant_weight_golden(case):
    from dipoorlet_amd.weight_transform.ada_quant_layer import RoundingParam, quant_weight
    k = case["key"]
    w, mask, G, scale = dev(Z[k + "_w"]), dev(Z[k + "_mask"]), dev(Z[k + "_G"]), dev(Z[k + "_scale"])
    qmin, qmax = torch.full_like(scale, -127.0), torch.full_like(scale, 127.0)
    close(quant_weight(w, mask, scale, qmin, qmax, case["per_channel"], soft=True), Z[k + "_soft"])
    assert np.array_equal(quant_weight(w, mask, scale, qmin, qmax, case["per_channel"], soft=False).cpu().numpy(),
                          Z[k + "_hard"])
    rp = RoundingParam(w, scale, qmin, qmax, case["per_channel"])
    close(rp.round_mask, Z[k + "_alpha0"], rtol=1e-5, atol=2e-6)
    if not case["tight"]:     # nothing clamps: h(alpha0) gives the weight back
        close(rp.qw, Z[k + "_w"], rtol=0, atol=float(scale.max()) * 2e-6 * 127)
    rp.round_mask.copy_(mask)
    g, _ = mask_gradient(rp, G)
    close(g, Z[k + "_grad"], rtol=5e-6, atol=1e-9)


def test_regulariser_golden():
    from dipoorlet_amd.weight_transform.ada_quant_layer import TempDecay, adaround_reg
    mask = dev(Z["reg_mask"])
    close(adaround_reg().rectified_sigmoid(mask), Z["rect_sigmoid"])
    for t, v in META["temp_decay_1000"].items():
        assert TempDecay(1000)(int(t)) == pytest.approx(v, abs=1e-12)
    for row in META["reg"]:
        reg = adaround_reg(row["max_iter"])
        val, g = reg.value_and_grad(mask, row["iter"])
        assert reg.beta == pytest.approx(row["beta"], abs=1e-9)
        assert float(val) == pytest.approx(row["value"], rel=2e-5, abs=1e-6)
        close(g, Z[f"reg_grad_{row['max_iter']}_{row['iter']}"], rtol=2e-4, atol=2e-8)


def test_l2_norm_golden_and_relu():
    from dipoorlet_amd.weight_transform.ada_quant_layer import L2_norm
    from oracle import round_oracle as ro
    for row in META["l2"]:
        k = row["key"]
        val, g = L2_norm(dev(Z[k + "_pred"]), dev(Z[k + "_tgt"]))
        assert float(val) == pytest.approx(row["value"], rel=1e-6)
        close(g, Z[k + "_grad"])
    rng = np.random.default_rng(3)
    for shape in ((5, 7, 9, 11), (3, 1000), (2, 3, 5)):          # ragged sizes: the scalar tail of the vector kernel
        p, t = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
        for relu in (False, True):
            ev, eg = ro.l2_value_grad(p, t, relu)
            val, g = L2_norm(dev(p), dev(t), relu=relu)
            assert float(val) == pytest.approx(ev, rel=1e-6)
            close(g, eg)
    # accumulation into a caller buffer and a non-16-B-aligned view
    buf = torch.zeros(1, dtype=torch.float64, device="cuda")
    p = dev(rng.standard_normal(4099).astype(np.float32))[1:].reshape(2, -1)
    t = torch.zeros_like(p)
    L2_norm(p, t, loss=buf)
    L2_norm(p, t, loss=buf)
    assert float(buf) == pytest.approx(2 * float((p.double() ** 2).sum()) / 2, rel=1e-9)


def test_quant_acti_drop_golden_and_autograd():
    from dipoorlet_amd.weight_transform.ada_quant_layer import quant_acti
    d = META["drop"][0]
    x, r, G = dev(Z["drop_x"]), dev(Z["drop_r"]), dev(Z["drop_G"])
    y = quant_acti(x, d["scale"], d["q_min"], d["q_max"], d["prob"], rand=r)
    assert np.array_equal(y.cpu().numpy(), Z["drop_y"])
    # the autograd path draws its own uniform numbers: the gradient is 1 exactly where the value was kept
    xa = x.clone().requires_grad_(True)
    torch.manual_seed(0)
    ya = quant_acti(xa, d["scale"], d["q_min"], d["q_max"], 0.5)
    (ya * G).sum().backward()
    kept = (ya.detach() == x) & (ya.detach() != quant_acti(x, d["scale"], d["q_min"], d["q_max"], 1.0))
    quantised = ya.detach() != x
    assert torch.equal(xa.grad[kept], G[kept]) and float(xa.grad[quantised].abs().sum()) == 0.0
    assert 0.35 < float(quantised.float().mean()) < 0.65
    # prob = 1: everything quantised, no gradient
    xb = x.clone().requires_grad_(True)
    quant_acti(xb, d["scale"], d["q_min"], d["q_max"], 1.0).sum().backward()
    assert float(xb.grad.abs().sum()) == 0.0


def test_step_matches_oracle_on_more_seeds():
    """One fused step (gradient + regulariser + Adam + refreshed weight) against the hand-written oracle."""
    from dipoorlet_amd.weight_transform.ada_quant_layer import RoundingParam
    from oracle import round_oracle as ro
    rng = np.random.default_rng(11)
    for shape, pc in (((16, 8, 3, 3), True), ((16, 8, 3, 3), False), ((10, 33), True), ((1, 7), False)):
        w = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        amax = np.abs(w).reshape(shape[0], -1).max(1) if pc else np.abs(w).max(keepdims=True).reshape(1)
        scale = (amax / 100.0).astype(np.float32)             # < 127: some channels clamp
        qmin, qmax = np.full_like(scale, -127.0), np.full_like(scale, 127.0)
        rp = RoundingParam(dev(w), dev(scale), dev(qmin), dev(qmax), pc)
        _, mask = ro.alpha_init(w, scale)
        opt = ro.Adam(mask.shape)
        for it, beta in enumerate((0.0, 20.0, 7.5)):
            G = rng.standard_normal(shape).astype(np.float32)
            _, dq = ro.quant_weight(w, mask, scale, qmin, qmax, pc)
            rv, rg = ro.reg_value_grad(mask, beta)
            mask = opt.step(mask, (G * dq).astype(np.float32) + rg)
            rp.qw.grad = dev(G)
            regbuf = torch.zeros(1, dtype=torch.float64, device="cuda")
            rp.step(beta, reg_loss=regbuf)
            assert float(regbuf) == pytest.approx(rv, rel=3e-5, abs=1e-6)
            diff = np.abs(rp.round_mask.cpu().numpy() - mask)
            assert np.mean(diff < 3e-6) > 0.995 and diff.max() <= 2.1e-3 * (it + 1), (shape, pc, it, diff.max())
            mask = rp.round_mask.cpu().numpy().copy()          # re-synchronise: compare step by step
            opt.m, opt.v = rp.exp_avg.cpu().numpy().copy(), rp.exp_avg_sq.cpu().numpy().copy()
            close(rp.qw, ro.quant_weight(w, mask, scale, qmin, qmax, pc)[0], rtol=3e-6, atol=1e-8)
        assert np.array_equal(rp.hard_weight().cpu().numpy(), ro.quant_weight(w, mask, scale, qmin, qmax, pc, soft=False)[0])


def _node(kind):
    from dipoorlet_amd.onnx_io import Node
    if kind == "gemm":
        return Node("Gemm", ["x", "w", "b"], ["y"], name="fc", attrs={"transB": 1})
    return Node("Conv", ["x", "w", "b"], ["y"], name="conv", attrs={"pads": [1, 1, 1, 1], "kernel_shape": [3, 3],
                                                                     "strides": [1, 1], "dilations": [1, 1], "group": 1})


@pytest.mark.parametrize("row", META["traj"], ids=lambda r: r["key"])
def test_training_trajectory_golden(row):
    """learning_round_mask on the GPU against the reference's CPU trajectory."""
    from dipoorlet_amd.weight_transform.ada_quant_layer import AdaQLayer, adaround_reg
    from dipoorlet_amd.weight_transform.reconstruction import learn_rounding
    k = row["key"]
    scale = dev(Z[k + "_scale"])
    qw = {"scale": scale, "q_min": torch.full_like(scale, row["q_min"]), "q_max": torch.full_like(scale, row["q_max"]),
          "per_channel": row["per_channel"], "type": "Linear"}
    layer = AdaQLayer(_node(row["kind"]), dev(Z[k + "_w"]), dev(Z[k + "_b"]), qw, None, row["relu"], False)
    snaps = {}

    def grab(it, layers):
        if it in (1, 10, row["total_iter"]):
            snaps[it] = layers[0].round_mask.cpu().numpy().copy()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    learn_rounding([layer], dev(Z[k + "_x"]), None, dev(Z[k + "_fp"]), adaround_reg(row["total_iter"]), row["bs"],
                   row["epochs"], on_step=grab)
    for step, tol in ((1, 2e-6), (10, 5e-5), (row["total_iter"], 1e-3)):
        diff = np.abs(snaps[step] - Z[f"{k}_mask_{step}"])
        assert np.mean(diff <= tol) >= 0.95, (step, float(np.mean(diff <= tol)), float(diff.max()))
        assert diff.max() <= 2.5e-3 * step
    assert np.mean(layer.new_weight().cpu().numpy() == Z[k + "_hard"]) >= 0.97
    assert layer.rp.steps == row["total_iter"]


def test_graph_replay_equals_eager():
    """The hipGraph-replayed loop and the eager loop run the same kernels: masks must agree to rounding noise of the
    library convolutions, and both must have advanced the device schedule identically."""
    from dipoorlet_amd.weight_transform.ada_quant_layer import AdaQLayer, adaround_reg
    from dipoorlet_amd.weight_transform.reconstruction import learn_rounding
    row = next(r for r in META["traj"] if r["kind"] == "conv")
    k = row["key"]
    res = {}
    for mode in (False, True):
        scale = dev(Z[k + "_scale"])
        qw = {"scale": scale, "q_min": torch.full_like(scale, row["q_min"]), "q_max": torch.full_like(scale, row["q_max"]),
              "per_channel": row["per_channel"], "type": "Linear"}
        layer = AdaQLayer(_node("conv"), dev(Z[k + "_w"]), dev(Z[k + "_b"]), qw, None, row["relu"], False)
        reg = adaround_reg(row["total_iter"])
        learn_rounding([layer], dev(Z[k + "_x"]), None, dev(Z[k + "_fp"]), reg, row["bs"], row["epochs"], use_graph=mode)
        res[mode] = (layer.round_mask.cpu().numpy().copy(), reg.beta, layer.rp.steps)
    assert res[True][1] == res[False][1] and res[True][2] == res[False][2] == row["total_iter"]
    diff = np.abs(res[True][0] - res[False][0])
    assert np.mean(diff <= 1e-5) >= 0.99 and diff.max() <= 2.5e-3 * row["total_iter"]
    assert np.mean(np.abs(res[True][0] - Z[f"{k}_mask_{row['total_iter']}"]) <= 1e-3) >= 0.95
