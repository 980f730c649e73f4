"""`--update_bn` — counterpart of dipoorlet/weight_transform/update_bn.py: re-estimate the running statistics of
every BatchNormalization node from the activations the FAKE-QUANTISED network feeds it.

Reference (:26-48): per BN node it rebuilds the quantised graph, pulls the node's input for all N images out of the
host-side ActivationCache and folds the per-image channel mean / std into the running values with momentum 0.9, one
image after the other (:12-17 — note it folds the per-image *std* into `running_var`).  Here the quantised graph is
walked once, node-major, with the whole calibration set resident in HBM; the per-image channel statistics of a BN
input are two torch reductions on the device and the momentum recurrence is evaluated in closed form.
"""
import numpy as np
import torch

from ..executor import GraphSession
from ..forward_net import load_input_batch
from ..graph import ONNXGraph
from ..quantize import quant_graph
from ..utils import logger
from .bias_correction import _Frontier

__all__ = ["update_bn", "update_bn_multipass", "fold_running_stats"]


def fold_running_stats(running_mean, running_var, means, stds, momentum=0.9):
    """update_bn.py:12-17 — running <- m * running + (1 - m) * stat_i for i = 0..n-1, in the arrays' own dtype
    (fp32: numpy keeps it, the python-float momentum is a weak scalar)."""
    running_mean = np.asarray(running_mean)
    running_var = np.asarray(running_var)
    means = np.asarray(means, running_mean.dtype)
    stds = np.asarray(stds, running_var.dtype)
    for i in range(len(means)):
        running_mean = momentum * running_mean + (1.0 - momentum) * means[i]
        running_var = momentum * running_var + (1.0 - momentum) * stds[i]
    return running_mean, running_var


@torch.no_grad()
def update_bn_multipass(graph, act_clip_val, weight_clip_val, args, recalibrate=True):
    """update_bn.py:26-48 -> (graph with updated BN statistics, re-calibrated act ranges, weight ranges);
    saved as update_bn_model.onnx.  recalibrate=False (weight_calibration: the re-calibration there is a
    collective over all ranks) returns the graph alone."""
    from ..tensor_cali import tensor_calibration
    clip_val = {k: [np.copy(v[0]), np.copy(v[1])] for k, v in {**act_clip_val, **weight_clip_val}.items()}
    graph_bn = ONNXGraph()
    graph_bn.copy_from(graph)
    graph_q, _ = quant_graph(graph_bn, clip_val, args)
    dev = torch.device("cuda", torch.cuda.current_device())
    s_q = GraphSession(graph_q, device=dev)
    chunk = int(getattr(args, "calib_batch", 16) or 16)
    N = args.data_num            # rank 0 walks all images, like the reference (ActivationCache(graph_q, args))
    bounds = [(i, min(i + chunk, N)) for i in range(0, N, chunk)]
    sizes = [j - i for i, j in bounds]
    shapes = {n: graph.get_tensor_shape(n) for n in graph.network_inputs}
    qf = _Frontier(s_q, graph_q)
    for n in graph.network_inputs:
        qf.env[n] = [load_input_batch(args.input_dir, [n], shapes, i, j, dev)[n] for i, j in bounds]
    for node in graph_q.graph.node:
        if node.name in s_q._folded:
            continue
        if node.op_type == "BatchNormalization":
            logger.info("Update BN for node: {}".format(node.name))
            means, stds = [], []
            for t in qf.env[node.input[0]]:                       # [b, C, ...]: per-image channel statistics
                x = t.double().flatten(2)
                means.append(x.mean(2))
                stds.append(x.std(2, unbiased=False))             # np.std: population
            means = torch.cat(means).cpu().numpy()
            stds = torch.cat(stds).cpu().numpy()
            mean_name, var_name = node.input[3], node.input[4]
            new_mean, new_var = fold_running_stats(graph_bn.get_initializer(mean_name),
                                                   graph_bn.get_initializer(var_name), means, stds)
            for g in (graph_bn, graph_q):
                g.set_initializer(mean_name, new_mean.astype(np.float32))
                g.set_initializer(var_name, new_var.astype(np.float32))
            s_q.set_const(mean_name, torch.from_numpy(new_mean.astype(np.float32)))
            s_q.set_const(var_name, torch.from_numpy(new_var.astype(np.float32)))
        qf.run(node, len(bounds), sizes)
    graph_bn.update_model()
    if getattr(args, "output_dir", None):
        graph_bn.output_dir = args.output_dir
        graph_bn.save_onnx_model("update_bn_model")
    if not recalibrate:
        return graph_bn
    act_clip_val, weight_clip_val = tensor_calibration(graph_bn, args)
    return graph_bn, act_clip_val, weight_clip_val


update_bn = update_bn_multipass
