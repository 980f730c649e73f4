"""Helpers shared by the weight transforms — counterpart of dipoorlet/weight_transform/utils.py:1-65."""
import numpy as np
import torch

from ..quantize import get_qnode_by_param

LEARNABLE_LAYER_TYPES = ["Conv", "Gemm", "ConvTranspose"]
__all__ = ["LEARNABLE_LAYER_TYPES", "follow_relu", "following_relu", "update_weight", "get_quant_tensor",
           "get_block_from_first"]


def follow_relu(graph, node):
    """utils.py:12-15 — is the node's only consumer a Relu?"""
    nxt = graph.get_tensor_consumer(node.output[0])
    return len(nxt) == 1 and not isinstance(nxt[0], str) and nxt[0].op_type == "Relu"


def following_relu(graph, node):
    """utils.py:18-22."""
    nxt = graph.get_tensor_consumer(node.output[0])
    assert nxt[0].op_type == "Relu"
    return nxt[0]


def update_weight(graph, weight_tensor, weight_name):
    """utils.py:25-27."""
    graph.set_initializer(weight_name, np.asarray(weight_tensor, dtype=np.float32))


def get_quant_tensor(tensor_shape, param, tensor_range, device=None):
    """utils.py:30-50 — (scale, q_min, q_max) as fp32 device vectors with one entry per output channel
    (per-channel platforms) or a single entry.  The caller's range is not modified."""
    rng = [np.copy(tensor_range[0]), np.copy(tensor_range[1])]
    qnode, q_min, q_max = get_qnode_by_param(param, "tmp", tensor_shape, rng)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

    def vec(v):
        return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(-1).copy()).to(dev)
    return vec(qnode.scale), vec(q_min), vec(q_max)


def get_block_from_first(graph, node, args):
    """utils.py:53-65 — the chain of up to three learnable layers that starts at `node`, linked through single
    consumers (a Relu in between is allowed)."""
    res = [node]
    while True:
        nxt = graph.get_tensor_consumer(node.output[0])
        if len(nxt) != 1 or isinstance(nxt[0], str) or nxt[0].op_type not in LEARNABLE_LAYER_TYPES + ["Relu"]:
            return res
        if nxt[0].op_type != "Relu":
            res.append(nxt[0])
            if len(res) == 3:
                return res
        node = nxt[0]
