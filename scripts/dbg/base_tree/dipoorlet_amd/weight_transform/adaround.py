"""AdaRound — counterpart of dipoorlet/weight_transform/adaround.py (driver :19-116, learner :119-144)."""
from .ada_quant_layer import adaround_reg
from .reconstruction import learn_rounding, reconstruct

__all__ = ["adaround", "learning_round_mask"]


def adaround(graph_ori, graph, act_clip_val, weight_clip_val, args):
    """adaround.py:19-116 — learn, layer by layer, whether each weight rounds up or down so that the layer fed
    with the quantised network's activations reproduces the full-precision output; returns (and, on rank 0, saves
    as adaround.onnx) the graph with the hard-rounded weights.  The caller keeps using the ORIGINAL ranges."""
    return reconstruct(graph_ori, graph, act_clip_val, weight_clip_val, args, blockwise=False, save_name="adaround")


def learning_round_mask(in_tensor, fp_out_tensor, ada_layer, reg, batch_size, max_epoch):
    """adaround.py:119-144 — `ada_layer`: an AdaQLayer; returns its learned round mask (device tensor)."""
    reg = reg if reg is not None else adaround_reg()
    learn_rounding([ada_layer], in_tensor, None, fp_out_tensor, reg, batch_size, max_epoch, drop=False, log_every=50)
    return ada_layer.round_mask
