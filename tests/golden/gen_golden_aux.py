#!/usr/bin/env python3
"""Golden vectors for the rows round 1 left unpinned, produced by the REFERENCE's own, unmodified code
imported from /root/reference under the SURVEY Appendix-A stubs (build container only):

  cos_similarity                          dipoorlet/utils.py:273-278          (incl. the `dot == 0` branch)
  update_conv_node_bias                   dipoorlet/weight_transform/bias_correction.py:9-31
                                          (Conv [N,1,C,H,W] stacks, Gemm [N,1,C] stacks, with and without an existing bias)
  reduce_profiling_res                    dipoorlet/utils.py:386-412          (world sizes 1, 2, 3; with / without layer files)
  quant_graph / insert_fake_quant_node /  dipoorlet/quantize.py:20-108        WHICH tensors get fake-quantised, in which order,
  insert_fake_quant_node_output                                               how node inputs are re-wired, new network outputs

Only seeds, hand-written inputs and the reference's OUTPUTS are stored (aux_level.json / aux_level.npz).
Run:  python tests/golden/gen_golden_aux.py
"""
import copy
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
from _cases import AUX_GRAPH, aux_cos_pair, aux_stack  # noqa: E402

REF = "/root/reference"


def import_reference():
    for m in ["onnx", "onnx.helper", "onnx.numpy_helper", "onnx.external_data_helper", "onnxruntime",
              "onnxruntime.quantization", "onnxruntime.quantization.onnx_quantizer",
              "onnxruntime.quantization.quant_utils", "onnxsim", "termcolor"]:
        sys.modules[m] = mock.MagicMock(name=m)
    sys.path.insert(0, REF)
    import dipoorlet.quantize as q
    import dipoorlet.utils as ut
    import dipoorlet.weight_transform.bias_correction as bc
    return q, ut, bc


# ------------------------------------------------------------------------------------------------ cos_similarity
def cos_level(ut):
    rows = []
    for i in range(7):
        a, b = aux_cos_pair(i)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = ut.cos_similarity(a, b)
        rows.append({"case": i, "shape": list(a.shape), "cos": float(np.float64(c)), "dtype": type(c).__name__})
    return rows


# ------------------------------------------------------------------------------------------------ update_conv_node_bias
class _Init:
    def __init__(self, name, arr):
        self.name, self.arr = name, arr


class _BiasGraph:
    """What update_conv_node_bias touches of an ONNXGraph (bias_correction.py:14-31)."""

    def __init__(self, nodes, inits):
        self.graph = types.SimpleNamespace(node=nodes)
        self.initializer = {k: [_Init(k, v), i] for i, (k, v) in enumerate(inits.items())}
        self.tensor_name_shape_map = {k: list(v.shape) for k, v in inits.items()}
        self.input = list(inits)
        self.set_calls = []

    def set_initializer(self, name, value, raw=True):
        self.set_calls.append(name)
        self.initializer[name] = [_Init(name, np.asarray(value)), len(self.initializer)]


def bias_level(bc):
    bc.numpy_helper.to_array = lambda t: t.arr
    arrays, rows = {}, []
    for i, (op, C, hw, has_bias, n) in enumerate([("Conv", 6, (5, 7), True, 4), ("Conv", 3, (4, 4), False, 3),
                                                  ("Gemm", 10, None, True, 5), ("Gemm", 4, None, False, 2)]):
        fp, qq = aux_stack(i, n, C, hw)
        node = types.SimpleNamespace(name=f"node{i}", op_type=op, input=["x", "w"] + (["b"] if has_bias else []), output=["y"])
        inits = {"w": np.zeros((C, 1), np.float32)}
        if has_bias:
            inits["b"] = (np.arange(C, dtype=np.float32) * np.float32(0.25) - np.float32(1.0))
        g = _BiasGraph([node], inits)
        bc.update_conv_node_bias(g, node, [x for x in fp], [x for x in qq])
        name = "b" if has_bias else node.name + "_bias"
        new = g.initializer[name][0].arr
        arrays[f"bias/{i}"] = np.asarray(new)
        rows.append({"case": i, "op": op, "C": C, "hw": list(hw) if hw else None, "has_bias": has_bias, "n": n,
                     "bias_name": name, "node_inputs_after": list(node.input), "dtype": str(np.asarray(new).dtype),
                     "graph_input_appended": g.input[-1]})
    return rows, arrays


# ------------------------------------------------------------------------------------------------ reduce_profiling_res
def profiling_level(ut):
    runs = []
    rng = np.random.default_rng(77)
    for world, model_type in [(1, None), (2, None), (3, None), (3, "unet")]:
        with tempfile.TemporaryDirectory() as od:
            per_rank = []
            for r in range(world):
                layer = {f"t{k}": float(rng.uniform(0.9, 1.0)) for k in range(4)}
                model = {f"out{k}": [float(rng.uniform(0.9, 1.0)), float(rng.uniform(0.8, 0.9))] for k in range(2)}
                per_rank.append({"layer": layer, "model": model})
                if model_type is None:
                    with open(os.path.join(od, f"layer_res.json.rank{r}"), "w") as f:
                        json.dump(layer, f, indent=4)
                with open(os.path.join(od, f"model_res.json.rank{r}"), "w") as f:
                    json.dump(model, f, indent=4)
            args = types.SimpleNamespace(output_dir=od, model_type=model_type)
            layer, model = ut.reduce_profiling_res(world, args)
        runs.append({"world": world, "model_type": model_type, "per_rank": per_rank, "layer": layer, "model": model})
    return runs


# ------------------------------------------------------------------------------------------------ quant_graph selection
class _RefGraph:
    """Stand-in for the reference's ONNXGraph: exactly the members quantize.py:20-108 uses, plus a log."""

    def __init__(self):
        self.graph = types.SimpleNamespace(node=[])
        self.initializer, self.network_inputs, self.network_outputs, self.input = {}, [], [], []
        self.shapes, self.output_map, self.name_idx_map = {}, {}, {}
        self.inserted = []

    @classmethod
    def build(cls, spec):
        g = cls()
        g.graph.node = [types.SimpleNamespace(name=n["name"], op_type=n["op"], input=list(n["in"]), output=list(n["out"]))
                        for n in spec["nodes"]]
        g.initializer = {k: None for k in spec["initializers"]}
        g.network_inputs = list(spec["inputs"])
        g.network_outputs = list(spec["outputs"])
        g.input = list(spec["inputs"]) + list(spec["initializers"])
        g.topologize_graph()
        g.set_index()
        return g

    def copy_from(self, src):
        self.__dict__.update(copy.deepcopy(src.__dict__))

    def get_tensor_shape(self, name):
        return [1]

    def topologize_graph(self):
        self.output_map = {o: n for n in self.graph.node for o in n.output}

    def get_tensor_producer(self, name):
        return self.output_map.get(name, "INPUT_TOKEN")

    def set_index(self):
        self.name_idx_map = {n.name: i for i, n in enumerate(self.graph.node)}

    def index(self, node):
        return self.name_idx_map[node.name]

    def insert_qnodes_purely(self, q_nodes, idx=0, node=None):
        if node:
            idx = self.index(node)
        for nd in reversed(q_nodes.node):
            self.graph.node.insert(idx, nd)
        self.inserted.append(q_nodes.tensor)
        self.set_index()

    def del_network_output(self, name):
        self.network_outputs.remove(name)

    def add_network_output(self, out):
        self.network_outputs.append(out.name)

    def update_model(self):
        self.set_index()


def selection_level(q):
    def fake_qdq(tensor_name, tensor_shape, scale, zp, need_transpose=False, per_channel=False, symmetric=True):
        qn = types.SimpleNamespace(name=tensor_name + "_QuantizeLinear", op_type="QuantizeLinear", input=[tensor_name],
                                   output=[tensor_name + "_q"])
        dq = types.SimpleNamespace(name=tensor_name + "_DequantizeLinear", op_type="DequantizeLinear",
                                   input=[tensor_name + "_q"], output=[tensor_name + "_dq"])
        return types.SimpleNamespace(node=[qn, dq], initializer=[], tensor=tensor_name,
                                     output=[types.SimpleNamespace(name=tensor_name + "_dq")],
                                     per_channel=bool(per_channel), symmetric=bool(symmetric), transpose=bool(need_transpose))
    q.make_quant_dequant = fake_qdq
    q.ONNXGraph = _RefGraph
    src = _RefGraph.build(AUX_GRAPH)
    clip = {t: [np.float64(-1.0 - 0.1 * i), np.float64(2.0 + 0.1 * i)] for i, t in enumerate(AUX_GRAPH["tensors"])}
    for k, c in AUX_GRAPH["initializers"].items():
        clip[k] = [-np.ones(c), np.ones(c)]
    runs = []
    for deploy, skip in [("trt", []), ("snpe", []), ("ti", []), ("atlas", []), ("rv", []), ("trt", ["conv3", "relu_b"])]:
        args = types.SimpleNamespace(deploy=deploy, skip_layers=skip)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gq, qlist = q.quant_graph(src, copy.deepcopy(clip), args)
        runs.append({"deploy": deploy, "skip_layers": skip, "quantized_in_order": list(gq.inserted),
                     "quant_node_list": [n.name for n in qlist],
                     "node_inputs": {n.name: list(n.input) for n in gq.graph.node if not n.name.endswith("Linear")},
                     "node_order": [n.name for n in gq.graph.node],
                     "network_outputs": list(gq.network_outputs)})
    return runs


if __name__ == "__main__":
    q, ut, bc = import_reference()
    bias_rows, bias_arrays = bias_level(bc)
    out = {"numpy": np.__version__, "cos": cos_level(ut), "bias": bias_rows, "profiling": profiling_level(ut),
           "selection": selection_level(q)}
    with open(os.path.join(HERE, "aux_level.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "aux_level.npz"), **bias_arrays)
    print("aux_level:", len(out["cos"]), "cos cases,", len(bias_rows), "bias cases,", len(out["profiling"]), "profiling runs,",
          len(out["selection"]), "selection runs")
