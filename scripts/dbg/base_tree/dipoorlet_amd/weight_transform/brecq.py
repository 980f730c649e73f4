"""BRECQ / QDrop — counterpart of dipoorlet/weight_transform/brecq.py (driver :20-155, learner :158-200)."""
from .reconstruction import learn_rounding, reconstruct

__all__ = ["brecq", "learning_round_mask"]


def brecq(graph_ori, graph, act_clip_val, weight_clip_val, args):
    """brecq.py:20-155 — like AdaRound but over blocks of up to three chained learnable layers reconstructed
    jointly; with args.drop (QDrop) the block input mixes quantised and full-precision activations per epoch and
    every layer output is fake-quantised with probability 0.5 per element.  Saved as brecq.onnx on rank 0."""
    return reconstruct(graph_ori, graph, act_clip_val, weight_clip_val, args, blockwise=True, save_name="brecq")


def learning_round_mask(q_in_tensor, fp_in_tensor, fp_out_tensor, ada_block, reg, batch_size, max_epoch, drop):
    """brecq.py:158-200 — `ada_block`: a sequence of AdaQLayers; returns their learned round masks."""
    layers = list(ada_block)
    learn_rounding(layers, q_in_tensor, fp_in_tensor, fp_out_tensor, reg, batch_size, max_epoch, drop=drop,
                   log_every=100)
    return [layer.round_mask for layer in layers]
