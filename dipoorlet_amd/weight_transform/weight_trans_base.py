"""weight_calibration — counterpart of dipoorlet/weight_transform/weight_trans_base.py:15-68: the order in which
the weight transforms run and what is re-derived after each."""
import os

import torch.distributed as dist

from ..graph import ONNXGraph
from ..tensor_cali import find_clip_val_minmax_weight
from ..utils import logger
from .adaround import adaround
from .bias_correction import bias_correction
from .brecq import brecq

NOT_BUILT = ("we", "update_bn", "sparse")


def weight_calibration(onnx_graph, act_clip_val, weight_clip_val, args):
    """Returns (graph_after_wt, onnx_graph, act_clip_val, weight_clip_val) like the reference; every rank ends
    with the same model and ranges."""
    for flag in NOT_BUILT:
        if getattr(args, flag, False):
            raise NotImplementedError(f"--{flag} (weight equalisation / BN re-estimation / sparse) is outside this "
                                      "package's scope; see DESIGN.md")
    graph_after_wt = ONNXGraph()
    graph_after_wt.copy_from(onnx_graph)
    if getattr(args, "bc", False):   # :21-29 — rank 0 corrects, everyone reloads, weight (bias) ranges refreshed
        if dist.get_rank() == 0:
            logger.info("Weight transform: bias correction...")
            bias_correction(graph_after_wt, act_clip_val, weight_clip_val, args)
        dist.barrier()
        args.model = os.path.join(args.output_dir, "update_bias_model.onnx")
        graph_after_wt = ONNXGraph.load(args.model, args.output_dir, args.deploy, getattr(args, "model_type", None))
        weight_clip_val = find_clip_val_minmax_weight(graph_after_wt, args)
    if getattr(args, "adaround", False):   # :55-57
        args.acti_quant = False
        graph_after_wt = adaround(onnx_graph, graph_after_wt, act_clip_val, weight_clip_val, args)
    if getattr(args, "brecq", False):      # :59-64
        args.acti_quant = bool(getattr(args, "drop", False))
        graph_after_wt = brecq(onnx_graph, graph_after_wt, act_clip_val, weight_clip_val, args)
    return graph_after_wt, onnx_graph, act_clip_val, weight_clip_val
