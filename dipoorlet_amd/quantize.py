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

    def apply(self, x, out=None):
        """Fake-quantise a device tensor: QuantizeLinear -> DequantizeLinear semantics, one kernel."""
        if self._dev is None or self._dev[0].device != x.device:
            self._dev = (torch.from_numpy(self.scale).to(x.device),
                         torch.from_numpy(self.zero_point_as_stored()).to(x.device))
        lo, hi = self.saturation()
        axis = self.axis if self.scale.size > 1 else None
        return ops.fake_quant(x, self._dev[0], self._dev[1], lo, hi, axis=axis, out=out)


def get_qnode_by_param(param, in_tensor_name, tensor_shape, range, need_transpose=False):
    """quantize.py:111-194 — returns (QDQNode, q_min, q_max).  `range` = [lo, hi] (scalars or per-channel
    arrays) and, like the reference, is collapsed / clamped IN PLACE when the platform quantises per
    tensor (:121-122) or asymmetrically per channel (:166-168)."""
    bit_width = param["bit_width"]
    zero_point = [0]
    per_channel = bool(param.get("per_channel", False))
    q_nodes = q_min = q_max = None
    if param["type"] == "Linear":
        symmetric = param["symmetric"]
        if not per_channel:
            range[0] = np.min(range[0])
            range[1] = np.max(range[1])
            if param.get("dynamic_sym", False) and np.abs(range[0] - 0.0) < 1e-6:
                symmetric = False  # one more bit for non-negative activations
        if symmetric:
            channel_num = len(range[0]) if isinstance(range[0], np.ndarray) else 1
            q_min = [-2 ** (bit_width - 1) + 1] * channel_num
            q_max = [2 ** (bit_width - 1) - 1] * channel_num
            data_max = np.max(np.abs(range), axis=0)
            scale = np.array(data_max) / q_max
            if np.any(scale == 0):
                scale = np.where(scale == 0, 1., scale)  # all-zero channel
            scale = scale.tolist()
        elif not isinstance(range[0], np.ndarray):
            data_min = min(0, range[0])
            data_max = max(0, range[1])
            scale = (data_max - data_min) / (2 ** bit_width - 1)
            if scale == 0.0:
                scale += 1.
            zero_point = np.round(-data_min / scale)
            q_min = [int(-zero_point)]
            q_max = [int(2 ** bit_width - 1 - zero_point)]
            scale = [float(scale)]
        else:
            data_min = range[0]
            data_min[data_min > 0.] = 0.
            data_max = range[1]
            data_max[data_max < 0.] = 0.
            scale = (data_max - data_min) / (2 ** bit_width - 1)
            if np.any(scale == 0):
                logger.warning("Find {} channels all zero in {}, set scale to 1.".format(
                    len(np.where(scale == 0)[0]), in_tensor_name))
                scale = np.where(scale == 0, 1., scale)
            zero_point = (-data_min / scale).round()
            q_min = (-zero_point).astype(np.int32).tolist()
            q_max = (2 ** bit_width - 1 - zero_point).astype(np.int32).tolist()
            scale = scale.tolist()
        if param.get("log_scale", False):
            scale = 2 ** np.round(np.log2(scale))
        scale = np.array(scale, dtype=np.float32)
        with np.errstate(all="ignore"):
            # np.full(shape, zp, dtype=np.int8) in the reference: values above 127 wrap (zp 191 -> -65)
            zp = np.broadcast_to(np.asarray(zero_point, np.float64), scale.shape)
            zero_point = ((zp.astype(np.int64) + 128) % 256 - 128).astype(np.int8)
        q_nodes = QDQNode(in_tensor_name, tensor_shape, scale, zero_point, need_transpose, per_channel, symmetric)
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
# Which tensors get fake-quantised: the reference's graph surgery (quantize.py:20-108) on this package's
# ONNXGraph.  Each inserted pair is one fused FakeQuant node (graph.insert_qnodes_purely).
def quant_graph(onnx_graph, clip_val, args):
    """quantize.py:20-37 — copy the graph, fake-quantise the inputs of every node whose op type is in the
    platform's quant_nodes (minus --skip_layers), optionally the network outputs."""
    from .graph import ONNXGraph
    from .platform_settings import platform_setting_table
    graph_q = ONNXGraph()
    graph_q.copy_from(onnx_graph)
    skip = getattr(args, "skip_layers", []) or []
    plat = platform_setting_table[args.deploy]
    quant_node_list = [n for n in graph_q.graph.node if n.name not in skip and n.op_type in plat["quant_nodes"]]
    act_quantized = []
    for node in quant_node_list:
        insert_fake_quant_node(graph_q, node, act_quantized, clip_val, args)
    if plat["quantize_network_output"]:
        insert_fake_quant_node_output(graph_q, clip_val, args)
    graph_q.update_model()
    return graph_q, quant_node_list


def insert_fake_quant_node(graph, node, act_quantized, data_range_list, args):
    """quantize.py:40-95 — per input of `node`: first initializer input of a weighted layer -> qw_params,
    later initializers -> qb_params if the platform has them, activations -> qi_params; a ReLU-type node
    directly behind Conv/Gemm/Eltwise/Add is left alone ("merge relu"); for TensorRT the first Conv-fed
    branch of an Add is left alone; a tensor already quantised is only re-wired."""
    from .platform_settings import LAYER_HAS_WEIGHT, platform_setting_table
    param = platform_setting_table[args.deploy]
    find_weight = False
    trt_merge_add = False
    for idx, in_tensor in enumerate(list(node.input)):
        if in_tensor == "":
            continue
        need_transpose = False
        shape = graph.tensor_name_shape_map.get(in_tensor)
        if node.op_type in RELU_TYPE:
            prev = graph.get_tensor_producer(node.input[0])
            if isinstance(prev, str):
                continue
            if len(node.input) == 1 and prev.op_type in MERGE_RELU:
                continue
        q_nodes = None
        if in_tensor in graph.initializer and node.op_type in LAYER_HAS_WEIGHT:
            if not find_weight:
                find_weight = True
                need_transpose = node.op_type == "ConvTranspose"
                q_nodes, _, _ = get_qnode_by_param(param["qw_params"], in_tensor, shape, data_range_list[in_tensor],
                                                   need_transpose)
            elif "qb_params" in param:
                q_nodes, _, _ = get_qnode_by_param(param["qb_params"], in_tensor, shape, data_range_list[in_tensor],
                                                   need_transpose)
        if in_tensor in graph.network_inputs or in_tensor not in graph.input:
            if args.deploy == "trt" and node.op_type == "Add" and not trt_merge_add:
                prev = graph.get_tensor_producer(in_tensor)
                if not isinstance(prev, str) and prev.op_type == "Conv":
                    trt_merge_add = True
                    continue
            q_nodes, _, _ = get_qnode_by_param(param["qi_params"], in_tensor, shape, data_range_list[in_tensor])
        if q_nodes is not None:
            node.input[idx] = q_nodes.output
            if in_tensor in act_quantized:
                continue
            graph.insert_qnodes_purely(q_nodes=q_nodes, node=node)
            act_quantized.append(in_tensor)
    graph.topologize_graph()


def insert_fake_quant_node_output(graph, clip_val, args):
    """quantize.py:98-108 — fake-quantise every network output and make `<out>_dq` the new output."""
    from .platform_settings import platform_setting_table
    param = platform_setting_table[args.deploy]
    for out_tensor in list(graph.network_outputs):
        q_nodes, _, _ = get_qnode_by_param(param["qi_params"], out_tensor, graph.tensor_name_shape_map.get(out_tensor),
                                           clip_val[out_tensor])
        graph.insert_qnodes_purely(q_nodes=q_nodes, idx=graph.index(graph.get_tensor_producer(out_tensor)) + 1)
        graph.del_network_output(out_tensor)
        graph.add_network_output(q_nodes.output)
    graph.topologize_graph()
