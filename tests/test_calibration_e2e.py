"""GPU: the registry algorithms end to end through the reference-shaped API
(tensor_cali_dispatcher(key, graph, args) -> {name: [lo, hi]}), .bin ingest included, against the
clip ranges the reference itself produced for the same calibration set (tests/golden/pipeline_level.json)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from _cases import MINI_NET, mini_net_activations

pytestmark = pytest.mark.gpu


class MiniSession:
    """Plays the network: returns the prescribed activations of whichever images are in the batch."""

    def __init__(self, device):
        self.tensor_names = [n for n, _, _ in MINI_NET]
        self.elems_per_image = [e for _, e, _ in MINI_NET]
        self.device = device
        self.by_key = {}
        for i in range(8):
            acts = mini_net_activations(i)
            self.by_key[float(acts[0][1][0])] = acts

    def run(self, inputs):
        x = inputs["input"]
        b = x.shape[0]
        flat = x.reshape(b, -1)
        keys = flat[:, 0].cpu().numpy()
        per = [self.by_key[float(k)] for k in keys]
        out = [x.reshape(b, -1).contiguous()]
        for t in range(1, len(MINI_NET)):
            out.append(torch.from_numpy(np.stack([p[t][1] for p in per])).to(self.device))
        return out


class MiniGraph:
    network_inputs = ["input"]

    def get_tensor_shape(self, name):
        return [1, 3, 32, 32]

    def make_session(self, args):
        return MiniSession(torch.device("cuda:0"))


@pytest.fixture(scope="module")
def calib_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("calib")
    os.makedirs(d / "input")
    for i in range(8):
        mini_net_activations(i)[0][1].tofile(d / "input" / f"{i}.bin")
    return str(d)


@pytest.fixture(scope="module")
def golden_runs(golden_dir):
    with open(os.path.join(golden_dir, "pipeline_level.json")) as f:
        return json.load(f)["runs"]


def _args(calib_dir, run, rank, **kw):
    a = types.SimpleNamespace(input_dir=calib_dir, data_num=8, rank=rank, local_rank=0, world_size=run["world_size"],
                              bins=run["bins"], threshold=run["threshold"], deploy=run["deploy"],
                              act_quant=run["algo"], optim_transformer=False, merge="reference", calib_batch=3)
    a.__dict__.update(kw)
    return a


def _check(algo, got, ref):
    for k, v in ref.items():
        g = [float(got[k][0]), float(got[k][1])]
        if algo == "mse":
            assert np.allclose(g, v, rtol=1e-5, atol=1e-5), (k, g, v)
        else:
            assert g == v, (algo, k, g, v)  # bit-exact (fp32 -> python float is exact)


def test_registry_matches_reference_per_rank(calib_dir, golden_runs):
    """Every golden run, every rank, with the reference's own multi-rank semantics (merge='reference')."""
    from dipoorlet_amd.tensor_cali import tensor_cali_dispatcher
    for run in golden_runs:
        for rank in range(run["world_size"]):
            args = _args(calib_dir, run, rank)
            got = tensor_cali_dispatcher(run["algo"], MiniGraph(), args)
            assert hasattr(got["conv1"][0], "tolist")  # numpy scalars, as save_clip_val needs
            _check(run["algo"], got, run["ranks"][rank])


def test_batch_size_does_not_change_results(calib_dir, golden_runs):
    from dipoorlet_amd.tensor_cali import tensor_cali_dispatcher
    run = [r for r in golden_runs if r["algo"] == "hist" and r["world_size"] == 1 and r["bins"] == 2048][0]
    for bsz in (1, 8, 5):
        got = tensor_cali_dispatcher("hist", MiniGraph(), _args(calib_dir, run, 0, calib_batch=bsz))
        _check("hist", got, run["ranks"][0])
    # no HBM budget: pass 2 re-runs the forward instead of re-reading resident activations
    got = tensor_cali_dispatcher("hist", MiniGraph(), _args(calib_dir, run, 0, resident_gb=0.0))
    _check("hist", got, run["ranks"][0])


def test_statistics_seam_shapes(calib_dir, golden_runs, golden_dir):
    from dipoorlet_amd import forward_net as fn
    st = np.load(os.path.join(golden_dir, "pipeline_stats.npz"))
    run = golden_runs[0]
    args = _args(calib_dir, run, 0, deploy="ti")
    mm = fn.forward_get_minmax(MiniGraph(), args, per_image=True)
    hs = fn.forward_get_hist(MiniGraph(), mm, args)
    oc = fn.forward_net_octav(MiniGraph(), args)
    for k, _, _ in MINI_NET:
        assert np.array_equal(np.array(mm[k]["min"]), st[f"{k}/min"]) and np.array_equal(np.array(mm[k]["max"]), st[f"{k}/max"])
        assert np.array_equal(np.stack(hs[k]).sum(0), st[f"{k}/hist"].sum(0))
        assert np.allclose(np.array(oc[k]["optimal_s"]), st[f"{k}/octav_s_ti"], rtol=1e-5, atol=1e-5, equal_nan=True)
        assert len(oc[k]["optimal_s"]) == 8 and len(mm[k]["min"]) == 8


def test_store_stats_hook_and_weight_ranges(golden_dir):
    from dipoorlet_amd.tensor_cali import find_clip_val_hist, find_clip_val_minmax_weight
    kl = np.load(os.path.join(golden_dir, "kernel_level.npz"))
    key = "relu_25088_7" if "relu_25088_7/minmax" in kl else [k for k in kl.files if k.endswith("/minmax")][0][:-7]
    mn, mx = kl[key + "/minmax"]
    h = kl[key + "/hist_b2048_s1.0"]
    args = types.SimpleNamespace(bins=2048, threshold=0.999)
    got = find_clip_val_hist(None, args, store_stats={"minmax": {"t": {"min": [mn], "max": [mx]}}, "hist": {"t": h}})
    assert np.array_equal(np.array(got["t"], np.float32).view(np.uint32),
                          kl[key + "/hist_b2048_s1.0_clip0.999"].view(np.uint32))
    g = np.load(os.path.join(golden_dir, "qparam_level.npz"))
    w = {k: g[f"w/{k}"] for k in ("conv.w", "conv.b", "deconv.w", "gemm.w", "bn.scalar")}
    nodes = [types.SimpleNamespace(op_type="Conv", input=["x", "conv.w", "conv.b"]),
             types.SimpleNamespace(op_type="ConvTranspose", input=["y", "deconv.w"]),
             types.SimpleNamespace(op_type="Relu", input=["z"]),
             types.SimpleNamespace(op_type="BatchNormalization", input=["z", "bn.scalar"]),
             types.SimpleNamespace(op_type="Gemm", input=["z", "gemm.w"])]
    graph = types.SimpleNamespace(graph=types.SimpleNamespace(node=nodes), get_initializer=lambda n: w[n])
    wc = find_clip_val_minmax_weight(graph, None)
    assert sorted(wc) == ["conv.b", "conv.w", "deconv.w", "gemm.w"]
    for k in wc:
        assert np.array_equal(wc[k][0], g[f"wmin/{k}"]) and np.array_equal(wc[k][1], g[f"wmax/{k}"])
