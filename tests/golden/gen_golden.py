#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own, unmodified
calibration arithmetic (imported from /root/reference, build container only).

The reference cannot be imported as-is here (onnx / onnxruntime / onnxsim / termcolor are not
installed), so those third-party modules are replaced by MagicMock stand-ins in sys.modules and the
ONNXRuntime session is replaced by a fake that returns prescribed activations.
Everything numerical that runs is reference code + numpy:

  forward_get_minmax / forward_get_hist / forward_net_octav   dipoorlet/forward_net.py:192-342
  find_clip_val_minmax / _hist / _octav / _minmax_weight       dipoorlet/tensor_cali/basic_algorithm.py:13-91
  get_qnode_by_param (scale / zero-point / q-range)            dipoorlet/quantize.py:111-194
  save_clip_val / reduce_clip_val / load_clip_val              dipoorlet/utils.py:313-368
  quant_acti                                                   dipoorlet/weight_transform/ada_quant_layer.py:28-36

Only inputs' seeds + the reference's OUTPUTS are written (npz/json); no reference source travels.
Run:  python tests/golden/gen_golden.py      (needs /root/reference; not run on the GPU box)
"""
import json
import os
import sys
import tempfile
import types
import warnings
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _cases import MINI_NET, SIZES, checksum, make_tensor, mini_net_activations  # noqa: E402

REF = "/root/reference"


def import_reference():
    for m in ["onnx", "onnx.helper", "onnx.numpy_helper", "onnx.external_data_helper", "onnxruntime",
              "onnxruntime.quantization", "onnxruntime.quantization.onnx_quantizer",
              "onnxruntime.quantization.quant_utils", "onnxsim", "termcolor"]:
        sys.modules[m] = mock.MagicMock(name=m)
    sys.path.insert(0, REF)
    import dipoorlet.forward_net as fn
    import dipoorlet.quantize as q
    import dipoorlet.tensor_cali.basic_algorithm as ba
    import dipoorlet.utils as ut
    from dipoorlet.weight_transform.ada_quant_layer import quant_acti
    fn.copy.deepcopy = lambda x: x
    fn.tqdm = lambda it, **k: it
    return fn, ba, q, ut, quant_acti


class _Out:
    def __init__(self, name):
        self.name = name


class FakeSession:
    """Stands in for ort.InferenceSession: yields the prescribed activations image by image."""
    provider = None  # callable(run_index) -> [(name, array)]
    names = ()

    def __init__(self, *a, **k):
        self.calls = 0

    def get_provider_options(self):
        return {"CUDAExecutionProvider": {}}

    def get_outputs(self):
        return [_Out(n) for n in FakeSession.names]

    def run(self, outputs, feeds):
        acts = dict(FakeSession.provider(self.calls))
        self.calls += 1
        return [acts[n] for n in outputs]


def fake_graph(inputs):
    g = types.SimpleNamespace()
    g.network_inputs = [n for n, _ in inputs]
    shapes = dict(inputs)
    g.get_tensor_shape = lambda n: shapes[n]
    g.model = types.SimpleNamespace(graph=types.SimpleNamespace(node=[], output=[]),
                                    SerializeToString=lambda: b"")
    return g


def mk_args(**kw):
    a = types.SimpleNamespace(local_rank=0, rank=0, world_size=1, data_num=1, input_dir=None, bins=2048,
                              threshold=0.99999, deploy="trt", optim_transformer=False, act_quant="hist",
                              output_dir=None)
    a.__dict__.update(kw)
    return a


def write_bins(root, name, arrays):
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for i, x in enumerate(arrays):
        x.astype(np.float32).tofile(os.path.join(d, f"{i}.bin"))


# ------------------------------------------------------------------------------------------------
def kernel_level(fn, ba):
    """Single tensor, single image: min/max, |x| histogram, OCTAV scale, percentile clip."""
    out = {}
    meta = []
    cases = []
    for kind in ("normal", "relu", "laplace", "uniform", "zeros", "spike", "edges", "neg_only", "tiny", "with_nan"):
        for n in SIZES:
            if kind in ("zeros", "neg_only", "tiny", "with_nan") and n > 25088:
                continue
            if kind in ("laplace", "uniform", "spike") and n in (2048, 150528):
                continue
            cases.append((kind, n, len(cases)))
    for kind, n, seed in cases:
        x = make_tensor(kind, n, seed)
        key = f"{kind}_{n}_{seed}"
        with tempfile.TemporaryDirectory() as td:
            write_bins(td, "input", [x])
            FakeSession.names = ()
            FakeSession.provider = lambda i: []
            g = fake_graph([("input", (1, n))])
            for deploy in ("trt", "ti"):
                args = mk_args(input_dir=td, deploy=deploy)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    oc = fn.forward_net_octav(g, args)
                out[f"{key}/octav_{deploy}"] = np.array(
                    [oc["input"]["optimal_s"][0], oc["input"]["min"][0], oc["input"]["max"][0]], np.float32)
            args = mk_args(input_dir=td)
            mm = fn.forward_get_minmax(g, args)
            out[f"{key}/minmax"] = np.array([mm["input"]["min"][0], mm["input"]["max"][0]], np.float32)
            if kind == "with_nan":  # the reference's histogram pass raises on a NaN range: record that, no vectors
                try:
                    fn.forward_get_hist(g, mm, mk_args(input_dir=td))
                    raised = False
                except ValueError:
                    raised = True
                meta.append({"key": key, "kind": kind, "n": n, "seed": seed, "crc": checksum(x), "hist_raises": raised})
                continue
            for bins in (2048, 1000):
                for scale in (1.0, 1.5):
                    st = {"input": {"max": [np.float32(mm["input"]["max"][0] * np.float32(scale))],
                                    "min": [np.float32(mm["input"]["min"][0] * np.float32(scale))]}}
                    args = mk_args(input_dir=td, bins=bins)
                    h = fn.forward_get_hist(g, st, args)["input"][0]
                    assert h.dtype == np.int64
                    tag = f"{key}/hist_b{bins}_s{scale}"
                    out[tag] = h
                    for thr in (0.99999, 0.999):
                        args = mk_args(input_dir=td, bins=bins, threshold=thr)
                        cv = ba.find_clip_val_hist(g, args, store_stats={"minmax": st, "hist": {"input": h}})
                        out[f"{tag}_clip{thr}"] = np.array(cv["input"], np.float32)
        meta.append({"key": key, "kind": kind, "n": n, "seed": seed, "crc": checksum(x)})
    np.savez_compressed(os.path.join(HERE, "kernel_level.npz"), **out)
    with open(os.path.join(HERE, "kernel_level.json"), "w") as f:
        json.dump({"numpy": np.__version__, "cases": meta}, f, indent=1)
    print("kernel_level:", len(meta), "cases,", len(out), "arrays")


# ------------------------------------------------------------------------------------------------
def pipeline_level(fn, ba, ut):
    """MINI_NET, N images, sharded over world_size ranks; the three registry algorithms end to end,
    plus the reference's JSON save / rank-0 reduce / load."""
    N = 8
    res = {"numpy": np.__version__, "N": N, "runs": []}
    inp_name, inp_n, _ = MINI_NET[0]
    with tempfile.TemporaryDirectory() as td:
        write_bins(td, inp_name, [mini_net_activations(i)[0][1] for i in range(N)])
        g = fake_graph([(inp_name, (1, inp_n))])
        FakeSession.names = tuple(n for n, _, _ in MINI_NET[1:])
        for algo, deploy, bins, thr, world in [
            ("minmax", "trt", 2048, 0.99999, 1), ("minmax", "trt", 2048, 0.99999, 2),
            ("hist", "trt", 2048, 0.99999, 1), ("hist", "trt", 2048, 0.99999, 2),
            ("hist", "snpe", 1000, 0.999, 1), ("hist", "trt", 2048, 0.99999, 3),
            ("mse", "trt", 2048, 0.99999, 1), ("mse", "ti", 2048, 0.99999, 1),
            ("mse", "trt", 2048, 0.99999, 2), ("mse", "ti", 2048, 0.99999, 4),
            ("minmax", "trt", 2048, 0.99999, 8), ("hist", "trt", 2048, 0.99999, 8), ("mse", "trt", 2048, 0.99999, 8),
        ]:
            run = {"algo": algo, "deploy": deploy, "bins": bins, "threshold": thr, "world_size": world,
                   "ranks": []}
            with tempfile.TemporaryDirectory() as od:
                for rank in range(world):
                    rank_num = N // world
                    st = rank * rank_num
                    FakeSession.provider = lambda i, st=st: mini_net_activations(st + i)[1:]
                    args = mk_args(input_dir=td, deploy=deploy, bins=bins, threshold=thr, rank=rank,
                                   world_size=world, data_num=N, act_quant=algo, output_dir=od)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        clip = ba.tensor_cali_dispatcher(algo, g, args)
                    run["ranks"].append({k: [float(np.float64(v[0])), float(np.float64(v[1]))]
                                         for k, v in clip.items()})
                    # np.float32 -> python float is exact, so the json holds the fp32 values bit for bit
                    ut.save_clip_val(clip, {}, args, act_fname=f"act_clip_val.json.rank{rank}",
                                     weight_fname=f"weight_clip_val.json.rank{rank}")
                args = mk_args(deploy=deploy, act_quant=algo, output_dir=od)
                ut.reduce_clip_val(world, args)
                with open(os.path.join(od, "act_clip_val.json")) as f:
                    run["merged_json_text"] = f.read()
                act, _ = ut.load_clip_val(args)
                run["merged"] = {k: [float(v[0]), float(v[1])] for k, v in act.items()}
            res["runs"].append(run)
        # per-image statistics (the dict-of-lists seam) for world_size 1
        FakeSession.provider = lambda i: mini_net_activations(i)[1:]
        args = mk_args(input_dir=td, data_num=N)
        mm = fn.forward_get_minmax(g, args)
        hs = fn.forward_get_hist(g, mm, args)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            oc = fn.forward_net_octav(g, mk_args(input_dir=td, data_num=N, deploy="ti"))
        stats = {}
        for k in mm:
            stats[f"{k}/min"] = np.array(mm[k]["min"], np.float32)
            stats[f"{k}/max"] = np.array(mm[k]["max"], np.float32)
            stats[f"{k}/hist"] = np.stack(hs[k]).astype(np.int64)
            stats[f"{k}/octav_s_ti"] = np.array(oc[k]["optimal_s"], np.float32)
        np.savez_compressed(os.path.join(HERE, "pipeline_stats.npz"), **stats)
    with open(os.path.join(HERE, "pipeline_level.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("pipeline_level:", len(res["runs"]), "runs")


# ------------------------------------------------------------------------------------------------
def qparam_level(q, ba, quant_acti):
    """scale / zero_point / q_min / q_max for every platform; weight per-channel min/max; quant_acti."""
    from dipoorlet.platform_settings import platform_setting_table
    captured = {}

    def capture(name, shape, scale, zp, need_transpose=False, per_channel=False, symmetric=True):
        captured.update(scale=np.array(scale), zp=np.array(zp), per_channel=bool(per_channel),
                        symmetric=bool(symmetric), need_transpose=bool(need_transpose))
        return "qnodes"
    q.make_quant_dequant = capture
    rng = np.random.default_rng(20240)
    ranges = [(-3.0, 1.0), (0.0, 6.0), (-1.25, 3.7), (0.0, 0.0), (-0.5, -0.1), (2e-7, 5.5), (-7.3, 7.1),
              (-300.0, 0.01)]
    rows = []
    for plat, tab in platform_setting_table.items():
        for pkey in ("qi_params", "qw_params"):
            param = tab[pkey]
            for lo, hi in ranges:
                r = [np.float64(lo), np.float64(hi)]
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    _, qmin, qmax = q.get_qnode_by_param(param, "t", [1], r)
                rows.append({"platform": plat, "param": pkey, "per_channel_in": False, "lo": [lo], "hi": [hi],
                             "scale": np.asarray(captured["scale"], np.float32).ravel().tolist(),
                             "zp": np.asarray(captured["zp"]).astype(np.int64).ravel().tolist(),
                             "qmin": np.asarray(qmin).ravel().tolist(), "qmax": np.asarray(qmax).ravel().tolist(),
                             "symmetric": captured["symmetric"], "per_channel": captured["per_channel"]})
            if pkey == "qw_params":
                for trial in range(3):
                    C = 5
                    lo = -np.abs(rng.standard_normal(C)) * (trial != 1)
                    hi = np.abs(rng.standard_normal(C))
                    if trial == 2:
                        lo[2] = 0.0
                        hi[2] = 0.0
                    r = [lo.copy(), hi.copy()]
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        _, qmin, qmax = q.get_qnode_by_param(param, "w", [C, 3], r)
                    rows.append({"platform": plat, "param": pkey, "per_channel_in": True,
                                 "lo": lo.tolist(), "hi": hi.tolist(),
                                 "scale": np.asarray(captured["scale"], np.float32).ravel().tolist(),
                                 "zp": np.asarray(captured["zp"]).astype(np.int64).ravel().tolist(),
                                 "qmin": np.asarray(qmin).ravel().tolist(),
                                 "qmax": np.asarray(qmax).ravel().tolist(),
                                 "symmetric": captured["symmetric"], "per_channel": captured["per_channel"]})
    # weight per-channel min/max (basic_algorithm.py:72-91)
    w = {"conv.w": rng.standard_normal((16, 3, 3, 3)).astype(np.float32),
         "conv.b": rng.standard_normal((16,)).astype(np.float32),
         "deconv.w": rng.standard_normal((4, 6, 2, 2)).astype(np.float32),
         "gemm.w": rng.standard_normal((10, 64)).astype(np.float32),
         "bn.scalar": np.float32(3.0)}
    nodes = [types.SimpleNamespace(op_type="Conv", input=["x", "conv.w", "conv.b"]),
             types.SimpleNamespace(op_type="ConvTranspose", input=["y", "deconv.w"]),
             types.SimpleNamespace(op_type="Relu", input=["z"]),
             types.SimpleNamespace(op_type="BatchNormalization", input=["z", "bn.scalar"]),
             types.SimpleNamespace(op_type="Gemm", input=["z", "gemm.w"])]
    g = types.SimpleNamespace(graph=types.SimpleNamespace(node=nodes),
                              get_initializer=lambda n: np.asarray(w[n]))
    wc = ba.find_clip_val_minmax_weight(g, None)
    arrays = {f"w/{k}": np.asarray(v) for k, v in w.items()}
    for k, (lo, hi) in wc.items():
        arrays[f"wmin/{k}"] = np.asarray(lo, np.float32)
        arrays[f"wmax/{k}"] = np.asarray(hi, np.float32)
    # torch-side fake quant (ada_quant_layer.py:28-36), prob = 1
    import torch
    x = make_tensor("normal", 4096, 5)
    x[:8] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 1e9, -1e9], np.float32) * np.float32(0.05)
    for i, (scale, qlo, qhi) in enumerate([(0.05, -127, 127), (0.031, -128, 127), (0.0123, 0, 255),
                                           (0.2, -7, 7)]):
        y = quant_acti(torch.from_numpy(x.copy()), torch.tensor(np.float32(scale)),
                       torch.tensor(float(qlo)), torch.tensor(float(qhi)), 1.0).numpy()
        arrays[f"qa/{i}/y"] = y
        arrays[f"qa/{i}/p"] = np.array([scale, qlo, qhi], np.float64)
    arrays["qa/x"] = x
    np.savez_compressed(os.path.join(HERE, "qparam_level.npz"), **arrays)
    with open(os.path.join(HERE, "qparam_level.json"), "w") as f:
        json.dump({"numpy": np.__version__, "rows": rows, "weight_keys": sorted(wc.keys())}, f, indent=1)
    print("qparam_level:", len(rows), "rows")


if __name__ == "__main__":
    fn, ba, q, ut, quant_acti = import_reference()
    fn.ort.InferenceSession = FakeSession
    kernel_level(fn, ba)
    pipeline_level(fn, ba, ut)
    qparam_level(q, ba, quant_acti)
