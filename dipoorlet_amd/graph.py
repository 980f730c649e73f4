"""Graph IR for calibration — the counterpart of the reference's ONNXGraph wrapper
(dipoorlet/utils.py:22-250), built on the package's own ONNX reader (onnx_io) because the `onnx`
package is not available.

Surface kept from the reference (what tensor_cali / quantize / forward_net touch):
    .graph.node (each with .name .op_type .input .output), .initializer (name -> ...), .network_inputs,
    .network_outputs, .input, .output, get_tensor_shape, get_initializer, set_initializer,
    get_tensor_producer ('INPUT_TOKEN' for graph inputs), get_tensor_consumer (['OUTPUT_TOKEN'] for
    leaves), topologize_graph, index, insert_node_purely, remove_node_purely, copy_from, update_model,
    save_onnx_model.
New: make_session(args) -> executor session exposing every node output (forward_net.py:193-202).
"""
import copy
import os
import types

import numpy as np

from . import onnx_io
from .onnx_io import Node


class ONNXGraph:
    def __init__(self, model=None, output_dir="", deploy=None, model_type=None):
        self.model = model
        self.output_dir, self.deploy, self.model_type = output_dir, deploy, model_type
        self.graph = types.SimpleNamespace(node=[], name="graph")
        self.initializer = {}           # name -> np.ndarray
        self.input_map, self.output_map = {}, {}
        self.network_inputs, self.network_outputs = [], []
        self.tensor_name_shape_map = {}
        self.name_idx_map = {}
        self.input, self.output = [], []
        self.opset = {"": 13}
        self.ir_version = 8
        self._qdq = {}                  # fused fake-quant nodes by node name (quantize.QDQNode)
        if model is not None:
            self._from_model(model)

    # ------------------------------------------------------------------ construction
    @classmethod
    def load(cls, path, output_dir="", deploy=None, model_type=None):
        return cls(onnx_io.load_model(path), output_dir, deploy, model_type)

    def _from_model(self, m):
        self.graph.node = list(m.nodes)
        self.graph.name = m.graph_name
        self.opset, self.ir_version = dict(m.opset), m.ir_version
        self.initializer = dict(m.initializers)
        for idx, node in enumerate(self.graph.node):          # set_names (utils.py:49-52)
            if node.name == "":
                node.name = node.op_type + "_" + str(idx)
        for node in self.graph.node:                           # convert_constant_to_init (:54-58)
            if node.op_type == "Constant" and "value" in node.attrs:
                self.initializer[node.output[0]] = np.asarray(node.attrs["value"])
        self.graph.node = [n for n in self.graph.node if n.op_type != "Constant"]
        self.topologize_graph()
        self.set_index()
        self._declared_inputs = list(m.inputs)
        self._declared_outputs = list(m.outputs)
        self._value_info = list(m.value_info)
        self.get_inp_oup()
        self.get_shape_type()

    def get_inp_oup(self):
        """utils.py:65-86."""
        self.network_inputs = [n for n, _, _ in self._declared_inputs
                               if isinstance(self.get_tensor_producer(n), str) and n not in self.initializer]
        self.network_outputs = [n for n, _, _ in self._declared_outputs]
        self.input = list(self.network_inputs)
        self.output = list(self.network_outputs)
        for node in self.graph.node:
            for inp in node.input:
                if inp in self.initializer and inp not in self.input:
                    self.input.append(inp)
            for oup in node.output:
                if oup not in self.output:
                    self.output.append(oup)

    def get_shape_type(self):
        """utils.py:88-117 (shapes only; '_q' / '_dq' aliases included)."""
        self.tensor_name_shape_map = {}
        for n, _, shp in self._declared_inputs:
            if n in self.network_inputs:
                self.tensor_name_shape_map[n] = list(shp or [])
        for n, _, shp in self._declared_outputs:
            self.tensor_name_shape_map[n] = list(shp or [])
        for n, arr in self.initializer.items():
            self.tensor_name_shape_map[n] = list(arr.shape)
        for n, _, shp in self._value_info:
            self.tensor_name_shape_map[n] = list(shp or [])
        for n in list(self.tensor_name_shape_map):
            self.tensor_name_shape_map[n + "_q"] = self.tensor_name_shape_map[n]
            self.tensor_name_shape_map[n + "_dq"] = self.tensor_name_shape_map[n]

    # ------------------------------------------------------------------ queries
    def get_tensor_shape(self, tensor_name):
        return self.tensor_name_shape_map[tensor_name]

    def set_tensor_shape(self, tensor_name, shape):
        for suffix in ("", "_q", "_dq"):
            self.tensor_name_shape_map[tensor_name + suffix] = list(shape)

    def get_initializer(self, initializer_name):
        return self.initializer[initializer_name]

    def set_initializer(self, initializer_name, value_tensor, raw=True):
        self.initializer[initializer_name] = np.asarray(value_tensor)
        self.tensor_name_shape_map[initializer_name] = list(np.asarray(value_tensor).shape)

    def del_initializer(self, initializer_name):
        self.initializer.pop(initializer_name, None)

    def topologize_graph(self):
        self.input_map, self.output_map = {}, {}
        for node in self.graph.node:
            for o in node.output:
                self.output_map[o] = node
            for i in node.input:
                self.input_map.setdefault(i, []).append(node)

    def get_tensor_producer(self, output_name):
        return self.output_map.get(output_name, "INPUT_TOKEN")

    def get_tensor_consumer(self, input_name):
        return self.input_map.get(input_name, ["OUTPUT_TOKEN"])

    def set_index(self):
        self.name_idx_map = {n.name: i for i, n in enumerate(self.graph.node)}

    def index(self, node):
        return self.name_idx_map[node.name]

    # ------------------------------------------------------------------ editing
    def remove_node_purely(self, node):
        self.graph.node.remove(node)

    def insert_node_purely(self, node, idx=0):
        self.graph.node.insert(idx, node)

    def insert_qnodes_purely(self, q_nodes, idx=0, node=None):
        """utils.py:198-206 — the reference inserts a QuantizeLinear + DequantizeLinear pair and their two
        initializers; here the pair is ONE fused 'FakeQuant' node executed by k_fake_quant (quantize.QDQNode)."""
        if node is not None:
            idx = self.index(node)
        fq = Node("FakeQuant", [q_nodes.tensor_name], [q_nodes.output], name=q_nodes.q_name)
        self._qdq[fq.name] = q_nodes
        self.graph.node.insert(idx, fq)
        self.initializer[q_nodes.scale_name] = q_nodes.scale if q_nodes.scale.size > 1 else q_nodes.scale.reshape(())
        zp = q_nodes.zero_point if q_nodes.symmetric else q_nodes.zero_point.view(np.uint8)
        self.initializer[q_nodes.zero_point_name] = zp if zp.size > 1 else zp.reshape(())
        self.set_index()

    def del_network_output(self, out_name):
        self.network_outputs.remove(out_name)

    def add_network_output(self, out_name):
        self.network_outputs.append(out_name if isinstance(out_name, str) else out_name.name)

    def update_model(self):
        self.set_index()
        self.topologize_graph()

    def copy_from(self, source_graph):
        for k, v in source_graph.__dict__.items():
            if k == "model":
                self.model = v
            elif k == "initializer":
                self.initializer = dict(v)  # arrays are replaced, never mutated in place
            else:
                setattr(self, k, copy.deepcopy(v))

    # ------------------------------------------------------------------ simplification
    def fold_batchnorm(self):
        """The one rewrite of `onnxsim.simplify` (dipoorlet/__main__.py:101) that changes what gets calibrated:
        a BatchNormalization whose input is produced by a Conv / ConvTranspose / Gemm with constant weights, and
        consumed by nothing else, is folded into that layer (W' = W * g / sqrt(var + eps) per output channel,
        b' = (b - mean) * g / sqrt(var + eps) + beta).  Returns the number of folded nodes."""
        folded = 0
        for bn in [n for n in self.graph.node if n.op_type == "BatchNormalization"]:
            prev = self.get_tensor_producer(bn.input[0])
            if isinstance(prev, str) or prev.op_type not in ("Conv", "ConvTranspose", "Gemm"):
                continue
            if len(self.get_tensor_consumer(bn.input[0])) != 1 or bn.input[0] in self.network_outputs:
                continue
            if prev.input[1] not in self.initializer or any(i not in self.initializer for i in bn.input[1:5]):
                continue
            if prev.op_type == "Gemm" and (not prev.attrs.get("transB", 0) or prev.attrs.get("alpha", 1.0) != 1.0
                                           or prev.attrs.get("beta", 1.0) != 1.0):
                continue
            gamma, beta, mean, var = (np.asarray(self.initializer[i], np.float64) for i in bn.input[1:5])
            k = gamma / np.sqrt(var + float(bn.attrs.get("epsilon", 1e-5)))
            w = np.asarray(self.initializer[prev.input[1]], np.float64)
            if prev.op_type == "ConvTranspose":      # [C_in, C_out / group, ...]: output channels on axis 1
                group = int(prev.attrs.get("group", 1))
                if group != 1:
                    continue
                w_new = w * k.reshape((1, -1) + (1,) * (w.ndim - 2))
            else:
                w_new = w * k.reshape((-1,) + (1,) * (w.ndim - 1))
            has_bias = len(prev.input) > 2 and prev.input[2] != ""
            b = np.asarray(self.initializer[prev.input[2]], np.float64) if has_bias else np.zeros_like(mean)
            b_new = (b - mean) * k + beta
            self.set_initializer(prev.input[1], w_new.astype(np.float32))
            bname = prev.input[2] if has_bias else prev.name + "_bias"
            self.set_initializer(bname, b_new.astype(np.float32))
            if not has_bias:
                prev.input = list(prev.input[:2]) + [bname]
                if bname not in self.input:
                    self.input.append(bname)
            prev.output[0] = bn.output[0]            # the layer now produces what the BN produced
            self.remove_node_purely(bn)
            folded += 1
        if folded:
            used = {i for n in self.graph.node for i in n.input}
            for name in [k for k in self.initializer if k not in used]:
                self.del_initializer(name)
            self.update_model()
        return folded

    # ------------------------------------------------------------------ I/O
    def to_model(self, expand_fake_quant=True):
        m = onnx_io.Model()
        m.ir_version, m.opset, m.graph_name = self.ir_version, dict(self.opset), self.graph.name
        nodes = []
        for n in self.graph.node:
            if n.op_type == "FakeQuant" and expand_fake_quant:   # emit the reference's 2-node form (quantize.py:208-231)
                q = self._qdq[n.name]
                attrs = {"axis": q.axis} if q.per_channel else {}
                nodes.append(Node("QuantizeLinear", [q.tensor_name, q.scale_name, q.zero_point_name], [q.q_output],
                                  name=q.q_name, attrs=attrs))
                nodes.append(Node("DequantizeLinear", [q.q_output, q.scale_name, q.zero_point_name], [q.output],
                                  name=q.dq_name, attrs=attrs))
            else:
                nodes.append(n)
        m.nodes = nodes
        m.initializers = dict(self.initializer)
        m.inputs = [(n, onnx_io.FLOAT, self.tensor_name_shape_map.get(n)) for n in self.network_inputs]
        m.outputs = [(n, onnx_io.FLOAT, self.tensor_name_shape_map.get(n)) for n in self.network_outputs]
        skip = set(self.network_inputs) | set(self.network_outputs) | set(self.initializer)
        m.value_info = [(n, onnx_io.FLOAT, s) for n, s in self.tensor_name_shape_map.items()
                        if n in self.output_map and n not in skip]
        return m

    def save_onnx_model(self, name="tmp", size_threshold=2048):
        path = name if name.endswith(".onnx") else os.path.join(self.output_dir, f"{name}.onnx")
        onnx_io.save_model(self.to_model(), path)
        return path

    # ------------------------------------------------------------------ execution
    def make_session(self, args=None, device=None, first_batch=None):
        from .executor import GraphSession
        return GraphSession(self, device=device, first_batch=first_batch)
