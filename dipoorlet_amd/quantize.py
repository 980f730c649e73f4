"""Scale / zero-point derivation and the fake-quant forward — counterpart of dipoorlet/quantize.py.

The reference derives (scale, zero_point, q_min, q_max) on the host (quantize.py:111-194), wraps them
in a QuantizeLinear -> DequantizeLinear ONNX sub-graph (:197-239) and lets ONNXRuntime execute it.
Here the derivation is the same host arithmetic (numpy float64 -> float32, bit-exact against the
reference-generated golden rows) and the Q->DQ pair is ONE fused HIP kernel (k_fake_quant_*): 4 B read
+ 4 B written per element instead of a quantised intermediate tensor.

Naming follows the reference so the emitted graph / deploy files stay drop-in:
    <t>_scale, <t>_zero_point, <t>_q, <t>_dq, nodes <t>_QuantizeLinear / <t>_DequantizeLinear.
"""
import numpy as np
import torch

from . import ops
from .utils import logger

QTENSORSUFFIX = "_q"
DQTENSORSUFFIX = "_dq"
QUANT_NODE_NAME_LIST = ["QuantizeLinear", "DequantizeLinear"]
MERGE_RELU = ["Conv", "Gemm", "Eltwise", "Add"]
RELU_TYPE = ["Relu", "PRelu", "Mul"]


class QDQNode:
    """The fused fake-quant stand-in for the reference's 2-node `graph_quant` (quantize.py:197-239)."""

    def __init__(self, tensor_name, tensor_shape, scale, zero_point, need_transpose, per_channel, symmetric):
        self.tensor_name = tensor_name
        self.tensor_shape = list(tensor_shape) if tensor_shape is not None else None
        self.scale = np.asarray(scale, np.float32).reshape(-1)
        self.zero_point = np.asarray(zero_point, np.int8).reshape(-1)  # stored through int8 like :185
        self.per_channel = bool(per_channel)
        self.symmetric = bool(symmetric)
        self.axis = (1 if need_transpose else 0) if per_channel else None  # :214, :220
        self.zp_dtype = "int8" if symmetric else "uint8"                  # :205-206
        self.q_name = tensor_name + "_QuantizeLinear"
        self.dq_name = tensor_name + "_DequantizeLinear"
        self.scale_name = tensor_name + "_scale"
        self.zero_point_name = tensor_name + "_zero_point"
        self.q_output = tensor_name + QTENSORSUFFIX
        self.output = tensor_name + DQTENSORSUFFIX
        self._dev = None

    def zero_point_as_stored(self):
        """The integers an ONNX runtime sees: int8 values, or the same bytes read as uint8."""
        return self.zero_point.astype(np.int32) if self.symmetric else self.zero_point.view(np.uint8).astype(np.int32)

    def saturation(self):
        """QuantizeLinear saturates to the zero-point dtype's full range (ONNX opset 13) — note: -128,
        not the q_min = -127 the reference computes at :134 for its torch-side code."""
        return (-128, 127) if self.symmetric else (0, 255)

    def apply(self, x, out=None, pre=None, x2=None):
        """Fake-quantise a device tensor: QuantizeLinear -> DequantizeLinear semantics, one kernel (pre / x2: the producer's
        ReLU or Add + ReLU fused in, ops.fake_quant)."""
        if self._dev is None or self._dev[0].device != x.device:
            self._dev = (torch.from_numpy(self.scale).to(x.device),
                         torch.from_numpy(self.zero_point_as_stored()).to(x.device))
        lo, hi = self.saturation()
        axis = self.axis if self.scale.size > 1 else None
        return ops.fake_quant(x, self._dev[0], self._dev[1], lo, hi, axis=axis, out=out, pre=pre, x2=x2)


def _int8_wrap(zero_point, shape):
    """The reference stores the zero point through np.full(..., dtype=np.int8) (quantize.py:185): values above 127 wrap
    (191 -> -65) and are later re-read as uint8 for asymmetric grids.  Reproduced on purpose."""
    with np.errstate(all="ignore"):
        z = np.broadcast_to(np.asarray(zero_point, np.float64), shape).astype(np.int64)
    return ((z + 128) % 256 - 128).astype(np.int8)


def _symmetric_grid(bits, lo, hi):
    """scale = max(|lo|, |hi|) / (2^(b-1) - 1) per channel, grid [-(2^(b-1) - 1), 2^(b-1) - 1], zero point 0."""
    top = 2 ** (bits - 1) - 1
    n = lo.size if isinstance(lo, np.ndarray) else 1
    q_hi = [top] * n
    scale = np.asarray(np.max(np.abs([lo, hi]), axis=0)) / q_hi
    scale = np.where(scale == 0, 1.0, scale)          # an all-zero channel quantises with scale 1
    return scale.tolist(), [0], [-top] * n, q_hi


def _affine_grid(bits, lo, hi, tensor_name):
    """Asymmetric grid over [min(lo, 0), max(hi, 0)]: scale = span / (2^b - 1), zp = round(-lo / scale),
    q range [-zp, 2^b - 1 - zp].  Per-channel bounds are clamped IN PLACE, as the reference's caller-visible arrays are."""
    levels = 2 ** bits - 1
    if isinstance(lo, np.ndarray):
        np.minimum(lo, 0.0, out=lo)
        np.maximum(hi, 0.0, out=hi)
        scale = (hi - lo) / levels
        dead = scale == 0
        if dead.any():
            logger.warning("%d all-zero channel(s) in %s: scale set to 1", int(dead.sum()), tensor_name)
            scale = np.where(dead, 1.0, scale)
        zp = (-lo / scale).round()
        return scale.tolist(), zp, (-zp).astype(np.int32).tolist(), (levels - zp).astype(np.int32).tolist()
    lo, hi = min(0, lo), max(0, hi)
    scale = (hi - lo) / levels
    if scale == 0.0:
        scale += 1.0
    zp = np.round(-lo / scale)
    return [float(scale)], zp, [int(-zp)], [int(levels - zp)]


def get_qnode_by_param(param, in_tensor_name, tensor_shape, range, need_transpose=False):
    """quantize.py:111-194 — returns (QDQNode, q_min, q_max) for the clip range `range` = [lo, hi] (scalars, or
    per-channel arrays for weights).  Kept from the reference because callers rely on it: a per-tensor platform
    collapses `range` to its overall min / max IN PLACE; `dynamic_sym` platforms switch a non-negative activation
    (|lo| < 1e-6) to the asymmetric grid — one more bit; `log_scale` snaps scales to powers of two."""
    if param["type"] != "Linear":
        return None, None, None
    per_channel = bool(param.get("per_channel", False))
    symmetric = param["symmetric"]
    if not per_channel:
        range[0], range[1] = np.min(range[0]), np.max(range[1])
        if param.get("dynamic_sym", False) and np.abs(range[0] - 0.0) < 1e-6:
            symmetric = False
    if symmetric:
        scale, zero_point, q_min, q_max = _symmetric_grid(param["bit_width"], range[0], range[1])
    else:
        scale, zero_point, q_min, q_max = _affine_grid(param["bit_width"], range[0], range[1], in_tensor_name)
    if param.get("log_scale", False):
        scale = 2 ** np.round(np.log2(scale))
    scale = np.array(scale, dtype=np.float32)
    q_nodes = QDQNode(in_tensor_name, tensor_shape, scale, _int8_wrap(zero_point, scale.shape), need_transpose, per_channel,
                      symmetric)
    return q_nodes, q_min, q_max


def quant_acti(x, scale, q_min, q_max, prob=1.0):
    """weight_transform/ada_quant_layer.py:28-36 on the device: round-half-even(x / scale), clamp to
    [q_min, q_max], * scale.  QDrop mixing (prob < 1) keeps the original value where rand >= prob."""
    sc = torch.as_tensor(scale, dtype=torch.float32, device=x.device).reshape(-1)
    zp = torch.zeros(sc.numel(), dtype=torch.int32, device=x.device)
    y = ops.fake_quant(x, sc, zp, int(q_min), int(q_max))
    if prob < 1.0:
        y = torch.where(torch.rand_like(x) < prob, y, x)
    return y


# ------------------------------------------------------------------------------------------------
# Which tensors get fake-quantised (quantize.py:20-108), on this package's ONNXGraph.  Each inserted pair is one fused
# FakeQuant node (graph.insert_qnodes_purely).  The reference decides input by input inside one loop; here the decision
# lives in a small rule object (`_NodeRules`) that the surgery consults input by input.
class _NodeRules:
    """Which platform parameter set quantises input `idx` of `node` — 'qw_params' (first constant input of a weighted
    layer), 'qb_params' (later constant inputs, only where the platform defines them), 'qi_params' (activations and
    network inputs) — or None.  Asked input by input, in order, WHILE the node is being re-wired, because two of the
    reference's rules read the live graph:
      * a ReLU-type node (Relu / PRelu / Mul) whose first input comes from outside the producer map, or a one-input one
        directly behind Conv / Gemm / Eltwise / Add, is left alone (the activation is merged into its producer).  The
        reference looks `node.input[0]` up again for every operand: once this pass has re-wired it to a fresh fake-quant
        output — not in the producer map until the node is finished — it reads as a graph input, so of Mul(a, b) only `a`
        is quantised unless `a` already had its fake-quant node (quantize.py:49-55; pinned by tests/golden/aux_level.json);
      * TensorRT: the first Conv-produced operand of an Add rides on the Conv's output scale (:80-84)."""

    def __init__(self, graph, node, plat, deploy):
        from .platform_settings import LAYER_HAS_WEIGHT
        self.graph, self.node, self.plat = graph, node, plat
        self.relu_like = node.op_type in RELU_TYPE
        self.weighted = node.op_type in LAYER_HAS_WEIGHT
        self.weight_seen = False
        self.trt_add_slot = deploy == "trt" and node.op_type == "Add"

    def role(self, name):
        graph, node = self.graph, self.node
        if name == "":
            return None
        if self.relu_like:
            producer = graph.get_tensor_producer(node.input[0])
            if isinstance(producer, str) or (len(node.input) == 1 and producer.op_type in MERGE_RELU):
                return None
        role = None
        if self.weighted and name in graph.initializer:
            if not self.weight_seen:
                self.weight_seen, role = True, "qw_params"
            elif "qb_params" in self.plat:
                role = "qb_params"
        if name in graph.network_inputs or name not in graph.input:
            producer = graph.get_tensor_producer(name)
            if self.trt_add_slot and not isinstance(producer, str) and producer.op_type == "Conv":
                self.trt_add_slot = False
                return None
            role = "qi_params"
        return role


def insert_fake_quant_node(graph, node, act_quantized, data_range_list, args):
    """quantize.py:40-95 for one node: derive the grid of every input that has a role, re-wire the input to the
    fake-quantised tensor, and insert the FakeQuant node unless the tensor already has one (`act_quantized`)."""
    from .platform_settings import platform_setting_table
    plat = platform_setting_table[args.deploy]
    rules = _NodeRules(graph, node, plat, args.deploy)
    for idx, name in enumerate(list(node.input)):
        role = rules.role(name)
        if role is None:
            continue
        transposed = role == "qw_params" and node.op_type == "ConvTranspose"
        q_nodes, _, _ = get_qnode_by_param(plat[role], name, graph.tensor_name_shape_map.get(name), data_range_list[name],
                                           transposed)
        if q_nodes is None:
            continue
        node.input[idx] = q_nodes.output
        if name not in act_quantized:
            graph.insert_qnodes_purely(q_nodes=q_nodes, node=node)
            act_quantized.append(name)
    graph.topologize_graph()


def insert_fake_quant_node_output(graph, clip_val, args):
    """quantize.py:98-108 — platforms that quantise the network outputs: `<out>_dq` replaces every `<out>`."""
    from .platform_settings import platform_setting_table
    qi = platform_setting_table[args.deploy]["qi_params"]
    for name in tuple(graph.network_outputs):
        q_nodes, _, _ = get_qnode_by_param(qi, name, graph.tensor_name_shape_map.get(name), clip_val[name])
        graph.insert_qnodes_purely(q_nodes=q_nodes, idx=graph.index(graph.get_tensor_producer(name)) + 1)
        graph.del_network_output(name)
        graph.add_network_output(q_nodes.output)
    graph.topologize_graph()


def quant_graph(onnx_graph, clip_val, args):
    """quantize.py:20-37 -> (fake-quantised copy of the graph, the nodes whose inputs were considered): every node whose
    op type is in the platform's `quant_nodes` and whose name is not in --skip_layers, in graph order; then the network
    outputs where the platform asks for it."""
    from .graph import ONNXGraph
    from .platform_settings import platform_setting_table
    plat = platform_setting_table[args.deploy]
    skipped = set(getattr(args, "skip_layers", None) or ())
    graph_q = ONNXGraph()
    graph_q.copy_from(onnx_graph)
    quant_node_list = [n for n in graph_q.graph.node if n.op_type in plat["quant_nodes"] and n.name not in skipped]
    done = []
    for node in quant_node_list:
        insert_fake_quant_node(graph_q, node, done, clip_val, args)
    if plat["quantize_network_output"]:
        insert_fake_quant_node_output(graph_q, clip_val, args)
    graph_q.update_model()
    return graph_q, quant_node_list
