"""GPU: `--sparse` (sparse + quantised weights) against reference-generated vectors (tests/golden/round_level.*:
prune masks, quant_weight_wo_roundmask with its straight-through gradient, an SGD + cosine-LR trajectory) and end to
end through the CLI."""
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


def test_prune_masks_golden():
    from dipoorlet_amd.weight_transform.sparse_quant_layer import create_nv24_mask, create_unstruction_mask, prune_weight
    w4, w2 = dev(Z["sp_w4"]), dev(Z["sp_w2"])
    assert np.array_equal(create_unstruction_mask(w4, 0.5).cpu().numpy(), Z["sp_mask_unstr_w4"])
    assert np.array_equal(create_unstruction_mask(w2, 0.3).cpu().numpy(), Z["sp_mask_unstr_w2_30"])
    assert np.array_equal(create_nv24_mask(w4, 2, 4).cpu().numpy(), Z["sp_mask_nv24_w4"])
    assert np.array_equal(create_nv24_mask(w2, 2, 4).cpu().numpy(), Z["sp_mask_nv24_w2"])
    assert float(create_unstruction_mask(w4, 0.0).min()) == 1.0
    p = prune_weight(w4, {"pattern": "nv24", "rate": 0.5}).cpu().numpy()
    assert np.array_equal(p, Z["sp_w4"] * Z["sp_mask_nv24_w4"])


@pytest.mark.parametrize("row", META["sparse_quant"], ids=lambda r: r["key"])
def test_quantiser_value_and_straight_through_gradient(row):
    from dipoorlet_amd import _hip
    from dipoorlet_amd.ops import _ptr, _stream
    from dipoorlet_amd.weight_transform.sparse_quant_layer import create_unstruction_mask, quant_weight_wo_roundmask
    k = row["key"]
    w, G, scale = dev(Z[row["w"]]), dev(Z[k + "_G"]), dev(Z[k + "_scale"])
    qmin, qmax = torch.full_like(scale, -127.0), torch.full_like(scale, 127.0)
    mask = create_unstruction_mask(w, 0.5)
    qw = quant_weight_wo_roundmask(w, scale, qmin, qmax, row["per_channel"], mask=mask)
    np.testing.assert_allclose(qw.cpu().numpy(), Z[k + "_qw"], rtol=1e-6, atol=1e-9)
    g = torch.empty_like(w)
    wc = w.clone()
    nch = scale.numel()
    _hip.check(_hip.lib().dpl_sparse_step(_ptr(G), _ptr(wc), _ptr(mask), None, _ptr(scale), _ptr(qmin), _ptr(qmax),
                                          w.numel(), nch, w.numel() // nch, 1 if row["per_channel"] else 0, 1.0, 0.0, 0.0,
                                          0.0, 1, 0, _ptr(g), _stream()), "dpl_sparse_step")
    np.testing.assert_allclose(g.cpu().numpy(), Z[k + "_grad"], rtol=2e-6, atol=1e-9)
    assert torch.equal(wc, w)                                      # update = 0: nothing written
    if row["per_channel"]:                                         # the clamp was hit and blocks the gradient there
        sat = np.abs(Z[k + "_qw"] / Z[k + "_scale"].reshape(-1, 1, 1, 1)) >= 127.0 - 1e-3
        assert sat.any() and np.all(Z[k + "_grad"][sat & (np.abs(Z[row["w"]]) > 1.28 * Z[k + "_scale"].reshape(-1, 1, 1, 1) * 100)] == 0)


def test_sgd_trajectory_golden():
    from dipoorlet_amd.onnx_io import Node
    from dipoorlet_amd.weight_transform.reconstruction import learn_sparse
    from dipoorlet_amd.weight_transform.sparse_quant_layer import SparseQLayer, cosine_lr
    t = META["sparse_traj"]
    node = Node("Conv", ["x", "w", "b"], ["y"], name="c", attrs={"pads": [1, 1, 1, 1], "kernel_shape": [3, 3],
                                                                  "strides": [1, 1], "dilations": [1, 1], "group": 1})
    scale = dev(Z["sptraj_scale"])
    qw = {"scale": scale, "q_min": torch.full_like(scale, -127.0), "q_max": torch.full_like(scale, 127.0),
          "per_channel": True}
    layer = SparseQLayer(node, dev(Z["sptraj_w"]), dev(Z["sptraj_b"]), qw, True, {"pattern": "unstruction", "rate": t["rate"]})
    torch.backends.cudnn.allow_tf32 = False
    last = learn_sparse(layer, dev(Z["sptraj_x"]), dev(Z["sptraj_fp"]), t["bs"], t["epochs"])
    assert last == pytest.approx(Z["sptraj_losses"][-1], rel=2e-3)
    np.testing.assert_allclose(layer.weight.cpu().numpy(), Z["sptraj_learned"], rtol=0, atol=2e-5)
    final = layer.new_weight().cpu().numpy()
    assert np.mean(final == Z["sptraj_final"]) >= 0.99 and np.mean(final == 0.0) >= 0.5
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3), T_max=12)
    for ep in range(12):
        assert cosine_lr(1e-3, ep, 12) == pytest.approx(sched.get_last_lr()[0], rel=1e-12, abs=1e-18)
        sched.optimizer.step()
        sched.step()


def test_sparse_cli(tmp_path):
    from dipoorlet_amd import models
    from dipoorlet_amd.__main__ import main
    from dipoorlet_amd.graph import ONNXGraph
    d = str(tmp_path)
    g = models.resnet18(seed=4, image=32)
    g.output_dir = d
    g.save_onnx_model("model")
    os.makedirs(os.path.join(d, "calib", "input"))
    rng = np.random.default_rng(1)
    for i in range(8):
        rng.standard_normal(3 * 32 * 32).astype(np.float32).tofile(os.path.join(d, "calib", "input", f"{i}.bin"))
    out = os.path.join(d, "out")
    assert main(["-M", os.path.join(d, "model.onnx"), "-I", os.path.join(d, "calib"), "-N", "8", "-A", "minmax", "-D",
                 "trt", "-O", out, "--calib_batch", "8", "--skip_profiling", "--sparse", "--pattern", "nv24", "--ada_bs",
                 "8", "--ada_epoch", "3"]) == 0
    g1 = ONNXGraph.load(os.path.join(out, "sparse_quant.onnx"))
    checked = 0
    for node in g1.graph.node:
        if node.op_type == "Conv" and g1.get_initializer(node.input[1]).shape[1] % 4 == 0:
            w = g1.get_initializer(node.input[1])
            grp = np.abs(w).transpose(0, 2, 3, 1).reshape(-1, 4)
            assert np.all((grp != 0).sum(1) <= 2), node.name        # 2 : 4 along the input channels
            checked += 1
    assert checked >= 15 and os.path.exists(os.path.join(out, "trt_clip_val.json"))
