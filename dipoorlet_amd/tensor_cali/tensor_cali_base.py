"""Entry point of the calibration phase (counterpart of dipoorlet/tensor_cali/tensor_cali_base.py:4-7)."""
from . import basic_algorithm as _algo


def tensor_calibration(onnx_graph, args):
    """-> (activation clip ranges from the algorithm registered under args.act_quant, per-channel weight ranges).
    Every rank calls this; the activation statistics are merged over ranks inside the algorithm.

    The reference walks the initializers for the weight ranges and then lets the algorithm build its ORT session
    (tensor_cali_base.py:5-6).  Here ONE CalibrationRun serves both: its reader thread starts on the .bin files, its session
    puts every initializer on the device once, the weight ranges are reduced from those resident tensors and the
    algorithm sweeps with the same session."""
    from ..forward_net import WALL, CalibrationRun, wall
    WALL.clear()      # (the host-wall breakdown --timing_json reports is this calibration's, not the process's)
    run = None
    if args.act_quant in _algo.tensor_cali_dispatcher.registry:
        run = CalibrationRun(onnx_graph, args)
    try:
        with wall("weight_ranges_s"):
            ranges = {"weight": _algo.find_clip_val_minmax_weight(onnx_graph, args, session=run.session if run else None)}
        with wall("activation_algorithm_s"):
            ranges["act"] = _algo.tensor_cali_dispatcher(args.act_quant, onnx_graph, args, run=run)
    finally:
        if run is not None:
            run.close()
    return ranges["act"], ranges["weight"]
