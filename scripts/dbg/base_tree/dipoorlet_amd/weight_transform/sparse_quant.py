"""`--sparse` — counterpart of dipoorlet/weight_transform/sparse_quant.py (driver :19-104, learner :107-130)."""
from .reconstruction import learn_sparse, reconstruct

__all__ = ["sparse_quant", "learning_sparse_quant"]


def sparse_quant(graph_ori, graph, act_clip_val, weight_clip_val, args):
    """sparse_quant.py:19-104 — layer by layer, learn the weight itself under a magnitude mask (--sparse_rate,
    --pattern unstruction | nv24) and the weight quantiser so that the layer fed with the quantised network's
    activations reproduces the full-precision output; saved as sparse_quant.onnx on rank 0."""
    return reconstruct(graph_ori, graph, act_clip_val, weight_clip_val, args, blockwise=False, save_name="sparse_quant",
                       sparse=True)


def learning_sparse_quant(in_tensor, fp_out_tensor, sparse_layer, batch_size, max_epoch):
    """sparse_quant.py:107-130 — `sparse_layer`: a SparseQLayer; returns its learned (dense, unquantised) weight."""
    learn_sparse(sparse_layer, in_tensor, fp_out_tensor, batch_size, max_epoch)
    return sparse_layer.weight
