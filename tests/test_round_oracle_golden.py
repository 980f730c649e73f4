"""CPU: oracle/round_oracle.py against the reference-generated vectors of tests/golden/round_level.*
(AdaRound / BRECQ / QDrop arithmetic and autograd gradients; generator: tests/golden/gen_golden_round.py)."""
import json
import os

import numpy as np
import pytest

from oracle import round_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "round_level.npz"))
META = json.load(open(os.path.join(HERE, "golden", "round_level.json")))


def close(a, b, rtol=2e-6, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", META["quant_weight"], ids=lambda c: c["key"])
def test_quant_weight_values_and_gradients(case):
    k = case["key"]
    w, mask, G, scale = Z[k + "_w"], Z[k + "_mask"], Z[k + "_G"], Z[k + "_scale"]
    qmin, qmax = np.full_like(scale, -127.0), np.full_like(scale, 127.0)
    soft, dq = ro.quant_weight(w, mask, scale, qmin, qmax, case["per_channel"], soft=True)
    hard, _ = ro.quant_weight(w, mask, scale, qmin, qmax, case["per_channel"], soft=False)
    close(soft, Z[k + "_soft"])
    assert np.array_equal(hard, Z[k + "_hard"])
    close(G * dq, Z[k + "_grad"], rtol=5e-6, atol=1e-9)
    _, a0 = ro.alpha_init(w, scale)
    close(a0, Z[k + "_alpha0"], rtol=1e-5, atol=2e-6)
    if case["tight"] and case["per_channel"]:
        assert np.any(np.abs(soft / ro._bc(scale, w.ndim)) >= 127.0 - 1e-3)    # the clamp was exercised


def test_temp_decay_and_regulariser():
    for t, v in META["temp_decay_1000"].items():
        assert ro.temp_decay(int(t), 1000) == pytest.approx(v, abs=1e-12)
    mask = Z["reg_mask"]
    close(ro.rect_sigmoid(mask)[0], Z["rect_sigmoid"])
    for row in META["reg"]:
        beta = ro.temp_decay(row["iter"], row["max_iter"])
        assert beta == pytest.approx(row["beta"], abs=1e-9)
        val, g = ro.reg_value_grad(mask, beta)
        assert val == pytest.approx(row["value"], rel=2e-5, abs=1e-6)
        close(g, Z[f"reg_grad_{row['max_iter']}_{row['iter']}"], rtol=2e-4, atol=2e-8)


def test_l2_norm():
    for row in META["l2"]:
        k = row["key"]
        val, g = ro.l2_value_grad(Z[k + "_pred"], Z[k + "_tgt"])
        assert val == pytest.approx(row["value"], rel=1e-6)
        close(g, Z[k + "_grad"])


def test_quant_acti_drop():
    d = META["drop"][0]
    y, dy = ro.quant_acti_drop(Z["drop_x"], Z["drop_r"], d["scale"], d["q_min"], d["q_max"], d["prob"])
    assert np.array_equal(y, Z["drop_y"])
    assert np.array_equal(dy * Z["drop_G"], Z["drop_grad"])


@pytest.mark.parametrize("row", META["traj"], ids=lambda r: r["key"])
def test_training_trajectory(row):
    k = row["key"]
    scale = Z[k + "_scale"]
    qmin, qmax = np.full_like(scale, row["q_min"]), np.full_like(scale, row["q_max"])
    mask, snaps, hard = ro.train_layer(row["kind"], Z[k + "_w"], Z[k + "_b"], Z[k + "_x"], Z[k + "_fp"], scale, qmin,
                                       qmax, row["per_channel"], row["relu"], row["bs"], row["epochs"],
                                       snapshots=(1, 10, row["total_iter"]))
    for step, tol in ((1, 2e-6), (10, 2e-5), (row["total_iter"], 5e-4)):
        diff = np.abs(snaps[step] - Z[f"{k}_mask_{step}"])
        # Adam divides by sqrt(v): an element whose gradient is rounding noise can step the other way (2 * lr)
        assert np.mean(diff <= tol) >= 0.97, (step, float(diff.max()))
        assert diff.max() <= 2.5e-3 * step
    assert np.mean(hard == Z[k + "_hard"]) >= 0.98


# ---- host-side weight transforms of the product (pure numpy: no GPU involved) against the same golden file
def test_weight_equalization_matches_reference():
    import types
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.onnx_io import Node
    from dipoorlet_amd.weight_transform.weight_equalization import (find_successor, node_has_equalized,
                                                                    weight_equalization)
    g = ONNXGraph()
    g.graph.node = [Node(op, i, o, name=n) for op, i, o, n in META["we"]["nodes"]]
    names = ("w1", "b1", "w2", "b2", "w3", "w4", "b4", "w5", "slope")
    g.initializer = {k: Z["we_in_" + k].copy() for k in names}
    g.update_model()
    nodes = {n.name: n for n in g.graph.node}
    assert [n.name for n in find_successor(nodes["conv1"], g)] == ["conv2"]
    assert node_has_equalized(g, nodes["conv3"]) and not node_has_equalized(g, nodes["conv4"])   # conv4 feeds conv5 AND add
    out = weight_equalization(g, types.SimpleNamespace(output_dir=None))
    for k in names:
        np.testing.assert_allclose(out.get_initializer(k), Z["we_out_" + k], rtol=2e-6, atol=1e-9, err_msg=k)
    assert not np.array_equal(Z["we_in_w2"], Z["we_out_w2"])            # something was equalised
    assert np.array_equal(g.get_initializer("w1"), Z["we_in_w1"])       # the caller's graph is untouched


def test_update_bn_running_statistics_match_reference():
    from dipoorlet_amd.weight_transform.update_bn import fold_running_stats
    x = Z["bn_x"]                                                        # [n, 1, C, H, W]
    means = np.stack([np.mean(t, axis=(0, 2, 3)) for t in x])
    stds = np.stack([np.std(t, axis=(0, 2, 3)) for t in x])
    m, v = fold_running_stats(Z["bn_mean0"], Z["bn_var0"], means, stds)
    assert m.dtype == np.float32 and v.dtype == np.float32 == np.dtype(META["bn"]["dtype_var"])
    np.testing.assert_array_equal(m, Z["bn_mean1"])
    np.testing.assert_array_equal(v, Z["bn_var1"])
