"""`--we` cross-layer weight equalisation — counterpart of dipoorlet/weight_transform/weight_equalization.py.

For every Conv whose only consumer chain is (Relu | PRelu)? -> Conv, the output channels of the first layer and the
matching input channels of the second are rescaled by s = r1 / sqrt(r1 * r2) (r = per-channel max |w|), repeated
until the weights stop moving (reference :38-94).  The reference walks the channels in Python; here one iteration
is a handful of vectorised fp32 array operations with the same elementwise arithmetic (weights are a few MB: this
is host work, no kernel).  Saved as weight_equal_model.onnx.
"""
import numpy as np

from ..graph import ONNXGraph
from ..utils import logger
from .utils import update_weight

__all__ = ["find_successor", "node_has_equalized", "weight_equalization", "converged"]


def find_successor(cur_node, graph):
    """weight_equalization.py:10-30 — the single Conv fed by this node directly or through one (P)Relu; any other
    consumer (or a graph output) disqualifies the node."""
    result = []
    for node in graph.get_tensor_consumer(cur_node.output[0]):
        if isinstance(node, str):
            return []
        if node.op_type in ("Relu", "PRelu"):
            for nxt in graph.get_tensor_consumer(node.output[0]):
                if not isinstance(nxt, str) and nxt.op_type == "Conv":
                    result.append(nxt)
                else:
                    return []
        elif node.op_type == "Conv":
            result.append(node)
        else:
            return []
    return result


def node_has_equalized(graph, node):
    """weight_equalization.py:33-35."""
    return len(find_successor(node, graph)) == 1


def converged(cur_weight, prev_weight, threshold=1e-4):
    """weight_equalization.py:97-101."""
    return (np.linalg.norm(cur_weight[0] - prev_weight[0]) + np.linalg.norm(cur_weight[1] - prev_weight[1])) < threshold


def _equalize_once(w1, w2, b1):
    """One pass of :60-80 over all groups and channels at once -> (new_w1, new_w2, new_b1)."""
    num_group = w1.shape[0] // w2.shape[1]
    ci, co = w1.shape[0] // num_group, w2.shape[0] // num_group     # channels per group: first layer out / second out
    r1 = np.abs(w1).reshape(w1.shape[0], -1).max(1).reshape(num_group, ci)
    w2g = w2.reshape((num_group, co) + w2.shape[1:])
    r2 = np.abs(w2g).max(axis=(1,) + tuple(range(3, w2g.ndim)))          # [group, in-channel]
    r1 = np.where(r1 < 1e-6, np.float32(0), r1)
    r2 = np.where(r2 < 1e-6, np.float32(0), r2)
    with np.errstate(all="ignore"):
        s = r1 / np.sqrt(r1 * r2)
    s = np.where(np.isinf(s) | np.isnan(s), np.float32(1.0), s).astype(w1.dtype)
    new_w1 = w1 / s.reshape((-1,) + (1,) * (w1.ndim - 1))
    new_w2 = (w2g * s.reshape((num_group, 1, ci) + (1,) * (w2g.ndim - 3))).reshape(w2.shape)
    new_b1 = None if b1 is None else b1 / s.reshape(-1)
    return new_w1.astype(w1.dtype), new_w2.astype(w2.dtype), new_b1


def weight_equalization(graph, args):
    """weight_equalization.py:38-94 -> the equalised graph (also saved as weight_equal_model.onnx)."""
    graph_we = ONNXGraph()
    graph_we.copy_from(graph)
    for node in graph_we.graph.node:
        if node.op_type != "Conv":
            continue
        succ = find_successor(node, graph_we)
        if len(succ) != 1:
            continue
        nxt = succ[0]
        it = 1
        while True:
            w1 = np.asarray(graph_we.get_initializer(node.input[1]))
            w2 = np.asarray(graph_we.get_initializer(nxt.input[1]))
            b1 = np.asarray(graph_we.get_initializer(node.input[2])) if len(node.input) == 3 else None
            logger.info("Cross Layer WE: {} --- {} Groups: {} Iter: {}".format(node.name, nxt.name,
                                                                               w1.shape[0] // w2.shape[1], it))
            n1, n2, nb = _equalize_once(w1, w2, b1)
            if converged([w1, w2], [n1, n2]):
                break                       # like the reference, the last (sub-threshold) update is not applied
            it += 1
            update_weight(graph_we, n1, node.input[1])
            update_weight(graph_we, n2, nxt.input[1])
            if nb is not None:
                update_weight(graph_we, nb, node.input[2])
    graph_we.update_model()
    if getattr(args, "output_dir", None):
        graph_we.output_dir = args.output_dir
        graph_we.save_onnx_model("weight_equal_model")
    return graph_we
