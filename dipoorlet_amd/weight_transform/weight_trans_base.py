"""weight_calibration — counterpart of dipoorlet/weight_transform/weight_trans_base.py:15-68: the order in which
the weight transforms run and what is re-derived after each."""
import os

import torch.distributed as dist

from ..graph import ONNXGraph
from ..tensor_cali import find_clip_val_minmax_weight, tensor_calibration
from ..utils import load_clip_val, logger, reduce_clip_val, save_clip_val
from .adaround import adaround
from .bias_correction import bias_correction
from .brecq import brecq
from .sparse_quant import sparse_quant
from .update_bn import update_bn
from .weight_equalization import weight_equalization



def _reload(name, args):
    args.model = os.path.join(args.output_dir, name + ".onnx")        # utils.update_model_path
    return ONNXGraph.load(args.model, args.output_dir, args.deploy, getattr(args, "model_type", None))


def _recalibrate(graph, args):
    """Re-derive the ranges of a changed model on every rank and pass them through the same per-rank files ->
    rank-0 reduce -> load sequence as __main__ (so all ranks end with identical, JSON-rounded values)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    act, weight = tensor_calibration(graph, args)
    save_clip_val(act, weight, args, act_fname=f"act_clip_val.json.rank{rank}",
                  weight_fname=f"weight_clip_val.json.rank{rank}")
    dist.barrier()
    if rank == 0:   # (as __main__: with the statistics merged over RCCL every rank already holds the whole set's clips)
        reduce_clip_val(world, args, already_merged=(getattr(args, "merge", "allreduce") != "reference"))
    dist.barrier()
    return load_clip_val(args)


def weight_calibration(onnx_graph, act_clip_val, weight_clip_val, args):
    """Returns (graph_after_wt, onnx_graph, act_clip_val, weight_clip_val) like the reference; every rank ends
    with the same model and ranges."""
    graph_after_wt = ONNXGraph()
    graph_after_wt.copy_from(onnx_graph)
    if getattr(args, "bc", False):   # :21-29 — the reference: rank 0 corrects, everyone reloads, weight (bias) ranges refreshed
        # here every rank corrects over its shard of the images and the per-channel sums are all-reduced (bias_correction);
        # --merge reference: rank 0 alone, over all images.  Rank 0 writes the model, everyone reloads it, as before.
        sharded = dist.get_world_size() > 1 and getattr(args, "merge", "allreduce") != "reference"
        if dist.get_rank() == 0:
            logger.info("Weight transform: bias correction...")
        if sharded or dist.get_rank() == 0:
            bias_correction(graph_after_wt, act_clip_val, weight_clip_val, args)
        dist.barrier()
        graph_after_wt = _reload("update_bias_model", args)
        weight_clip_val = find_clip_val_minmax_weight(graph_after_wt, args)
    if getattr(args, "we", False):   # :31-38 — equalise on rank 0, everyone reloads and re-calibrates
        if dist.get_rank() == 0:
            logger.info("Weight transform: cross-layer equalisation...")
            weight_equalization(graph_after_wt, args)
        dist.barrier()
        graph_after_wt = _reload("weight_equal_model", args)
        act_clip_val, weight_clip_val = _recalibrate(graph_after_wt, args)
    if getattr(args, "update_bn", False):   # :40-53
        if dist.get_rank() == 0:
            logger.info("Weight transform: BN statistics of the quantised network...")
            update_bn(graph_after_wt, act_clip_val, weight_clip_val, args, recalibrate=False)
        dist.barrier()
        graph_after_wt = _reload("update_bn_model", args)
        if dist.get_rank() == 0:
            logger.info("Re calibration...")
        act_clip_val, weight_clip_val = _recalibrate(graph_after_wt, args)
    if getattr(args, "sparse", False):     # :65-66 — instead of AdaRound / BRECQ
        args.acti_quant = False
        graph_after_wt = sparse_quant(onnx_graph, graph_after_wt, act_clip_val, weight_clip_val, args)
        return graph_after_wt, onnx_graph, act_clip_val, weight_clip_val
    if getattr(args, "adaround", False):   # :55-57
        args.acti_quant = False
        graph_after_wt = adaround(onnx_graph, graph_after_wt, act_clip_val, weight_clip_val, args)
    if getattr(args, "brecq", False):      # :59-64
        args.acti_quant = bool(getattr(args, "drop", False))
        graph_after_wt = brecq(onnx_graph, graph_after_wt, act_clip_val, weight_clip_val, args)
    return graph_after_wt, onnx_graph, act_clip_val, weight_clip_val
