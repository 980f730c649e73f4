"""Entry point of the calibration phase (counterpart of dipoorlet/tensor_cali/tensor_cali_base.py:4-7)."""
from . import basic_algorithm as _algo


def tensor_calibration(onnx_graph, args):
    """-> (activation clip ranges from the algorithm registered under args.act_quant, per-channel weight ranges).
    Every rank calls this; the activation statistics are merged over ranks inside the algorithm."""
    from ..forward_net import WALL, wall
    WALL.clear()      # (the host-wall breakdown --timing_json reports is this calibration's, not the process's)
    with wall("weight_ranges_s"):
        ranges = {"weight": _algo.find_clip_val_minmax_weight(onnx_graph, args)}
    with wall("activation_algorithm_s"):
        ranges["act"] = _algo.tensor_cali_dispatcher(args.act_quant, onnx_graph, args)
    return ranges["act"], ranges["weight"]
