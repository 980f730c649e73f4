"""Entry point of the calibration phase (dipoorlet/tensor_cali/tensor_cali_base.py:4-7): weight ranges,
then the activation algorithm selected by args.act_quant through the dispatcher."""
from .basic_algorithm import find_clip_val_minmax_weight, tensor_cali_dispatcher


def tensor_calibration(onnx_graph, args):
    weight_clip_val = find_clip_val_minmax_weight(onnx_graph, args)
    act_clip_val = tensor_cali_dispatcher(args.act_quant, onnx_graph, args)
    return act_clip_val, weight_clip_val
