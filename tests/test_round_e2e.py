"""GPU: AdaRound / BRECQ / QDrop end to end on a small ResNet-18 — through the CLI from a real .onnx + .bin
directory, and on two ranks (DDP-style gradient averaging over the process group).

To make rounding matter the weight grid is narrowed to 4 bits for these tests: learned rounding must then
reproduce the full-precision network clearly better than the rounded-to-nearest weights the calibration starts
from, every learned weight must sit on the grid next to the original value, and all ranks must hold the same model."""
import copy
import os
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

# (every test here re-runs or compares library forwards: deterministic algorithms, tests/conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.two_forwards]

N, IMG, BS, EPOCHS = 16, 32, 8, 60


def _make_workdir(d):
    from dipoorlet_amd import models
    g = models.resnet18(seed=3, image=IMG)
    g.output_dir = str(d)
    g.save_onnx_model("model")
    os.makedirs(os.path.join(d, "calib", "input"))
    rng = np.random.default_rng(9)
    for i in range(N):
        rng.standard_normal(3 * IMG * IMG).astype(np.float32).tofile(os.path.join(d, "calib", "input", f"{i}.bin"))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("round")
    _make_workdir(str(d))
    return d


@pytest.fixture()
def four_bit_weights():
    from dipoorlet_amd.platform_settings import platform_setting_table
    saved = copy.deepcopy(platform_setting_table["trt"])
    platform_setting_table["trt"]["qw_params"]["bit_width"] = 4
    yield
    platform_setting_table["trt"].clear()
    platform_setting_table["trt"].update(saved)


def _network_error(model_path, workdir, act_clip, args):
    """Mean squared error of the fake-quantised network's output against the full-precision network's."""
    from dipoorlet_amd.forward_net import load_input_batch
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
    g_fp = ONNXGraph.load(str(workdir / "model.onnx"))
    g = ONNXGraph.load(model_path)
    clip = {**act_clip, **find_clip_val_minmax_weight(g_fp, args)}   # "we must use original ranges" (adaround.py:115)
    clip = {k: [np.copy(v[0]), np.copy(v[1])] for k, v in clip.items()}
    gq, _ = quant_graph(g, clip, args)
    dev = torch.device("cuda:0")
    inp = load_input_batch(str(workdir / "calib"), g_fp.network_inputs, {"input": g_fp.get_tensor_shape("input")}, 0, N, dev)
    out_name = g_fp.network_outputs[0]
    fp = g_fp.make_session().run_named(inp, [out_name])[0]
    q = gq.make_session().run_named(inp, [gq.network_outputs[0]])[0]
    return float(((fp - q) ** 2).mean())


def _check_on_grid(model_path, workdir, args, bits=4):
    """Every learned weight = (floor(w / s) or floor(w / s) + 1) * s, inside the clamp range."""
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
    g0, g1 = ONNXGraph.load(str(workdir / "model.onnx")), ONNXGraph.load(model_path)
    wr = find_clip_val_minmax_weight(g0, args)
    qmax = 2 ** (bits - 1) - 1
    moved = 0
    for node in g0.graph.node:
        if node.op_type not in ("Conv", "Gemm"):
            continue
        w0, w1 = g0.get_initializer(node.input[1]), g1.get_initializer(node.input[1])
        lo, hi = wr[node.input[1]]
        scale = (np.maximum(np.abs(lo), np.abs(hi)) / qmax).astype(np.float32).reshape([-1] + [1] * (w0.ndim - 1))
        k = w1 / scale
        assert np.abs(k - np.rint(k)).max() < 1e-3, node.name
        assert np.rint(k).min() >= -qmax and np.rint(k).max() <= qmax
        up = np.rint(k) - np.floor(w0 / scale)
        assert set(np.unique(up)) <= {0.0, 1.0}, (node.name, np.unique(up))
        moved += int((np.rint(k) != np.rint(w0 / scale)).sum())
        if len(node.input) > 2:      # biases are untouched
            assert np.array_equal(g0.get_initializer(node.input[2]), g1.get_initializer(node.input[2]))
    return moved


def _cli(workdir, out, extra):
    from dipoorlet_amd.__main__ import main
    torch.manual_seed(1234)      # (mini-batch order and QDrop masks come from torch's generator: one trajectory per box, not one per run)
    rc = main(["-M", str(workdir / "model.onnx"), "-I", str(workdir / "calib"), "-N", str(N), "-A", "minmax", "-D",
               "trt", "-O", str(out), "--calib_batch", "8", "--skip_profiling", "--ada_bs", str(BS), "--ada_epoch",
               str(EPOCHS), *extra])
    assert rc == 0


def test_adaround_cli_beats_nearest_rounding(workdir, four_bit_weights):
    import json
    out = workdir / "out_ada"
    _cli(workdir, out, ["--adaround"])
    args = types.SimpleNamespace(deploy="trt", skip_layers=[])
    act = {k: [np.float64(v[0]), np.float64(v[1])] for k, v in json.load(open(out / "act_clip_val.json")).items()}
    moved = _check_on_grid(str(out / "adaround.onnx"), workdir, args)
    assert moved > 100          # it did not just reproduce round-to-nearest
    err_nearest = _network_error(str(workdir / "model.onnx"), workdir, act, args)
    err_ada = _network_error(str(out / "adaround.onnx"), workdir, act, args)
    print("adaround / nearest output error:", err_ada / err_nearest)
    assert err_ada < 0.8 * err_nearest, (err_ada, err_nearest)
    assert os.path.exists(out / "trt_clip_val.json")     # the run went on to deployment with the original ranges


def test_brecq_qdrop_cli(workdir, four_bit_weights):
    import json
    out = workdir / "out_brecq"
    _cli(workdir, out, ["--brecq", "--drop"])
    args = types.SimpleNamespace(deploy="trt", skip_layers=[])
    act = {k: [np.float64(v[0]), np.float64(v[1])] for k, v in json.load(open(out / "act_clip_val.json")).items()}
    assert _check_on_grid(str(out / "brecq.onnx"), workdir, args) > 100
    err_nearest = _network_error(str(workdir / "model.onnx"), workdir, act, args)
    err_brecq = _network_error(str(out / "brecq.onnx"), workdir, act, args)
    print("brecq+qdrop / nearest output error:", err_brecq / err_nearest)
    assert err_brecq < 0.9 * err_nearest, (err_brecq, err_nearest)
    log = open(out / "log.txt").read() if os.path.exists(out / "log.txt") else ""
    assert "Qdrop for:" in log or log == ""


def _worker(rank, world, port, wd):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DPL_DIST_BACKEND="gloo")
    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.platform_settings import platform_setting_table
    from dipoorlet_amd.tensor_cali import tensor_calibration
    from dipoorlet_amd.weight_transform import adaround
    platform_setting_table["trt"]["qw_params"]["bit_width"] = 4
    dist_helper.init_default()
    args = types.SimpleNamespace(model=os.path.join(wd, "model.onnx"), input_dir=os.path.join(wd, "calib"), data_num=N,
                                 rank=rank, local_rank=0, world_size=world, bins=2048, threshold=0.99999, deploy="trt",
                                 act_quant="minmax", optim_transformer=False, merge="allreduce", calib_batch=4,
                                 output_dir=os.path.join(wd, "out2"), skip_layers=[], ada_bs=4, ada_epoch=10,
                                 acti_quant=False, drop=False)
    os.makedirs(args.output_dir, exist_ok=True)
    g = ONNXGraph.load(args.model, args.output_dir, "trt")
    act, wt = tensor_calibration(g, args)
    g_ada = adaround(g, g, act, wt, args)
    np.savez(os.path.join(wd, f"weights{rank}.npz"), **{k: v for k, v in g_ada.initializer.items() if v.ndim >= 2})
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_learn_the_same_rounding(tmp_path):
    _make_workdir(str(tmp_path))
    port = 29900 + os.getpid() % 90
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "weights0.npz"), np.load(tmp_path / "weights1.npz")
    assert len(a.files) >= 21
    for k in a.files:       # identical models: gradients were averaged over the ranks, not rank-local
        assert np.array_equal(a[k], b[k]), k
    assert os.path.exists(tmp_path / "out2" / "adaround.onnx")
