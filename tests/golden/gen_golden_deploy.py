#!/usr/bin/env python3
"""Golden files for the rv / stpu deploy emitters: the reference's own emitters (dipoorlet/deploy/deploy_rv.py,
deploy_stpu.py; imported from /root/reference under the third-party stubs of gen_golden.py, build container only)
run on a small hand-made graph + clip ranges; their output files are stored verbatim as data
(tests/golden/deploy_level.json) next to the inputs that produced them.

Run:  python tests/golden/gen_golden_deploy.py
"""
import json
import os
import sys
import tempfile
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

NODES = [  # op, inputs, outputs, name, attrs
    ["Conv", ["input", "w1", "b1"], ["c1"], "conv1", {"group": 1, "kernel_shape": [3, 3], "strides": [1, 1]}],
    ["Relu", ["c1"], ["r1"], "relu1", {}],
    ["Conv", ["r1", "w2"], ["c2"], "conv2", {"group": 1, "kernel_shape": [3, 3], "strides": [2, 2]}],
    ["Conv", ["r1", "w3", "b3"], ["c3"], "conv3", {"group": 2, "kernel_shape": [1, 1], "strides": [1, 1]}],
    ["Concat", ["c2u", "c3"], ["cat"], "concat", {}],
    ["Upsample", ["c2"], ["c2u"], "up", {}],
    ["Conv", ["cat", "w4", "b4"], ["c4"], "conv4", {"group": 1, "kernel_shape": [3, 3], "strides": [1, 1]}],
    ["Sigmoid", ["c4"], ["s4"], "sig", {}],
    ["GlobalAveragePool", ["s4"], ["gap"], "gap", {}],
    ["Gemm", ["gap", "w5", "b5"], ["fc"], "fc", {}],
    ["Clip", ["fc"], ["output"], "clip", {}],
]
SHAPES = {"input": [1, 3, 16, 16], "w1": [8, 3, 3, 3], "w2": [6, 8, 3, 3], "w3": [6, 4, 1, 1], "w4": [10, 12, 3, 3],
          "w5": [5, 10], "gap": [1, 10, 1, 1]}


def make_inputs():
    rng = np.random.default_rng(77)
    order = ["input", "c1", "r1", "c2", "c3", "c2u", "cat", "c4", "s4", "gap", "fc", "output"]
    act = {}
    for i, t in enumerate(order):
        lo, hi = -abs(rng.normal()) * (1 + i), abs(rng.normal()) * (2 + i)
        if t in ("r1", "s4"):
            lo = 0.0
        act[t] = [float(lo), float(hi)]
    weights = {k: (rng.standard_normal(SHAPES[k]) * 0.2).astype(np.float32) for k in ("w1", "w2", "w3", "w4", "w5")}
    wclip = {k: [v.reshape(v.shape[0], -1).min(1).astype(np.float64), v.reshape(v.shape[0], -1).max(1).astype(np.float64)]
             for k, v in weights.items()}
    for b, c in (("b1", 8), ("b3", 6), ("b4", 10), ("b5", 5)):
        v = (rng.standard_normal(c) * 0.1)
        wclip[b] = [v.copy(), v.copy()]
    return act, wclip, weights


class _Node(types.SimpleNamespace):
    def get_attribute_value(self, name, default=None):
        return self.attrs.get(name, default)


class _Graph:
    def __init__(self, weights):
        self.graph = types.SimpleNamespace(node=[_Node(op_type=o, input=list(i), output=list(u), name=n, attrs=a)
                                                 for o, i, u, n, a in NODES])
        self.network_inputs = ["input"]
        self.initializer = {k: [v] for k, v in weights.items()}
        self.initializer.update({b: [np.zeros(1, np.float32)] for b in ("b1", "b3", "b4", "b5")})

    def get_tensor_consumer(self, t):
        r = [n for n in self.graph.node if t in n.input]
        return r if r else ["OUTPUT_TOKEN"]

    def get_tensor_producer(self, t):
        for n in self.graph.node:
            if t in n.output:
                return n
        return "INPUT_TOKEN"

    def get_tensor_shape(self, t):
        return SHAPES[t]


def main():
    for m in ["onnx", "onnx.helper", "onnx.numpy_helper", "onnx.external_data_helper", "onnxruntime",
              "onnxruntime.quantization", "onnxruntime.quantization.onnx_quantizer",
              "onnxruntime.quantization.quant_utils", "onnxsim", "termcolor"]:
        sys.modules[m] = mock.MagicMock(name=m)
    sys.path.insert(0, REF)
    import dipoorlet.deploy.deploy_rv as rv
    import dipoorlet.deploy.deploy_stpu as stpu
    stpu.numpy_helper.to_array = lambda t: np.asarray(t)
    out = {"nodes": NODES, "shapes": SHAPES, "files": {}}
    act, wclip, weights = make_inputs()
    out["act_clip"] = act
    out["weight_clip"] = {k: [np.asarray(v[0]).tolist(), np.asarray(v[1]).tolist()] for k, v in wclip.items()}
    out["weights"] = {k: v.tolist() for k, v in weights.items()}

    def clip():
        c = {k: [np.float64(v[0]), np.float64(v[1])] for k, v in act.items()}
        c.update({k: [np.array(v[0]), np.array(v[1])] for k, v in wclip.items()})
        return c
    import dipoorlet.deploy.deploy_atlas as atlas
    import dipoorlet.deploy.deploy_imx as imx
    import dipoorlet.deploy.deploy_magicmind as mm
    import dipoorlet.deploy.deploy_snpe as snpe
    import dipoorlet.deploy.deploy_ti as ti
    import dipoorlet.deploy.deploy_trt as trt

    def act_only():
        return {k: [np.float64(v[0]), np.float64(v[1])] for k, v in act.items()}
    emitters = (("rv", rv.gen_rv_yaml, False, clip), ("stpu", stpu.gen_stpu_minmax, False, clip),
                ("stpu_wg", stpu.gen_stpu_minmax, True, clip), ("trt", trt.gen_trt_range, False, act_only),
                ("snpe", snpe.gen_snpe_encodings, False, act_only), ("ti", ti.gen_ti_json, False, act_only),
                ("imx", imx.gen_imx_range, False, clip), ("magicmind", mm.gen_magicmind_proto, False, act_only),
                ("atlas", atlas.gen_atlas_quant_param, False, act_only))
    for tag, fn, wg, mk in emitters:
        with tempfile.TemporaryDirectory() as td:
            g = _Graph(weights)
            g.network_outputs = ["output"]
            fn(g, mk(), types.SimpleNamespace(output_dir=td, stpu_wg=wg, deploy=tag))
            for f in sorted(os.listdir(td)):
                out["files"][f"{tag}/{f}"] = open(os.path.join(td, f)).read()
    with open(os.path.join(HERE, "deploy_level.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote deploy_level.json:", sorted(out["files"]))


if __name__ == "__main__":
    main()
