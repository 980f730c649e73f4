"""Activation clip-range search: the three registry algorithms of the reference
(dipoorlet/tensor_cali/basic_algorithm.py:8-69) plus the per-channel weight ranges (:72-91), computed
on the MI355X.

Registry keys, call form and return type are the reference's: `tensor_cali_dispatcher(key, graph, args)`
-> {tensor_name: [lo, hi]} with numpy scalars (they must support .tolist(), utils.py:314-316); an
unknown key logs "Calibration Algorithm Not Found!" and returns None.

Multi-rank: with args.world_size > 1 and args.merge != 'reference' (default) the per-rank statistics
are merged with RCCL collectives (dist_helper.merge_*), so every rank returns the clip ranges of the
WHOLE calibration set — equal to the reference run with world_size = 1.  args.merge == 'reference'
keeps the reference's behaviour: per-rank ranges / histograms / means, to be averaged afterwards by
utils.reduce_clip_val exactly as __main__.py:121-128 does.
"""
import numpy as np
import torch

from .. import ops
from ..dist_helper import gather_rows, merge_hist, merge_ranges
from ..forward_net import (CalibrationRun, forward_get_minmax, forward_net_octav, hist_pass)
from ..platform_settings import LAYER_HAS_WEIGHT
from ..utils import dispatch_functool, logger


@dispatch_functool
def tensor_cali_dispatcher(*args, **kwargs):
    logger.info("Calibration Algorithm Not Found!")


def _merged(args):
    return getattr(args, "world_size", 1) > 1 and getattr(args, "merge", "allreduce") != "reference"


def _as_clip_dict(names, lo, hi):
    lo = lo.detach().cpu().numpy().astype(np.float32, copy=False)
    hi = hi.detach().cpu().numpy().astype(np.float32, copy=False)
    return {n: [lo[t], hi[t]] for t, n in enumerate(names)}


@tensor_cali_dispatcher.register("minmax")
def find_clip_val_minmax(onnx_graph, args, run=None, **kwargs):
    """basic_algorithm.py:13-22 — [min over images, max over images] per tensor.
    run: a CalibrationRun over (onnx_graph, args) to sweep with (tensor_calibration builds one); default: a new one."""
    run = run or CalibrationRun(onnx_graph, args)
    forward_get_minmax(onnx_graph, args, run=run)
    gmin, gmax = run.acc.gmin, run.acc.gmax
    if _merged(args):
        merge_ranges(gmin, gmax, args.world_size)
    return _as_clip_dict(run.names, gmin, gmax)


@tensor_cali_dispatcher.register("hist")
def find_clip_val_hist(onnx_graph, args, store_stats=None, run=None, **kwargs):
    """basic_algorithm.py:25-54 — percentile (cumulative mass >= args.threshold) of the |x| histogram.

    store_stats = {'minmax': {name: {'min': [...], 'max': [...]}}, 'hist': {name: int64[bins]}} skips the
    sweeps (the reference's unused hook, :26-29) and only runs the percentile search on the device."""
    bins = int(args.bins)  # the reference leaves a CLI --bins as str and crashes at :47; int() is the fix
    if store_stats:
        names = list(store_stats["hist"].keys())
        dev = torch.device("cuda", torch.cuda.current_device())
        acc = ops.CalibAccumulators(len(names), dev, bins)
        mm = store_stats["minmax"]
        acc.set_minmax(torch.tensor([float(np.min(mm[n]["min"])) for n in names], dtype=torch.float32, device=dev),
                       torch.tensor([float(np.max(mm[n]["max"])) for n in names], dtype=torch.float32, device=dev))
        acc.hist_prepare()
        acc.hist.copy_(torch.from_numpy(np.stack([np.asarray(store_stats["hist"][n], np.int64) for n in names])))
    else:
        run = run or CalibrationRun(onnx_graph, args)
        forward_get_minmax(onnx_graph, args, run=run, keep_resident=True)   # pass 1
        gmin, gmax = run.acc.gmin.clone(), run.acc.gmax.clone()
        if _merged(args):
            merge_ranges(gmin, gmax, args.world_size)
        acc = hist_pass(run, gmin, gmax, bins)                                # pass 2
        if _merged(args):
            merge_hist(acc.hist, args.world_size)
        names = run.names
    clip = acc.hist_percentile(float(args.threshold))
    return _as_clip_dict(names, clip[:, 0], clip[:, 1])


@tensor_cali_dispatcher.register("mse")
def find_clip_val_octav(onnx_graph, args, run=None, **kwargs):
    """basic_algorithm.py:57-69 — OCTAV scale per image, then [max(min_all, -mean s), min(max_all, mean s)].
    The mean is taken with numpy in fp32 over the per-image list exactly as the reference does, so it is
    reproduced bit for bit given the per-image scales (python max/min: a NaN mean falls back to the range)."""
    run = run or CalibrationRun(onnx_graph, args)
    forward_net_octav(onnx_graph, args, run=run, as_dict=False)
    rows = run.octav_rows
    if _merged(args):
        rows = gather_rows(rows, args.world_size)
    r = rows.detach().cpu().numpy().astype(np.float32, copy=False)
    # [3, T, n] contiguous: row [k, t] is what the reference's np.array(list_of_per_image_values) holds — a contiguous fp32
    # vector in image order — so numpy's mean / max / min over it are the reference's, bit for bit
    cols = np.ascontiguousarray(r.transpose(2, 1, 0))
    clip_val = {}
    for t, name in enumerate(run.names):
        with np.errstate(all="ignore"):
            mean_s = cols[0, t].mean()
        data_max = cols[2, t].max()
        data_min = cols[1, t].min()
        clip_val[name] = [max(data_min, -mean_s), min(data_max, mean_s)]
    return clip_val


def find_clip_val_minmax_weight(onnx_graph, args, session=None):
    """basic_algorithm.py:72-91 — per output channel [min, max] of every initializer input (node.input[1:])
    of Conv / Gemm / ConvTranspose / PRelu / BatchNormalization; ConvTranspose weights are viewed
    [1,0,2,3]-transposed; 0-d initializers are skipped.  Row reductions run in k_rowwise_minmax.

    session: an executor session of THIS graph — its initializers are on the device already (one transfer at session
    build) and are reduced where they lie; without one every initializer is uploaded here."""
    weight_tensor, need_transpose = {}, []
    for node in onnx_graph.graph.node:
        if node.op_type in LAYER_HAS_WEIGHT:
            for in_tensor in list(node.input)[1:]:
                weight_tensor[in_tensor] = onnx_graph.get_initializer(in_tensor)
            if node.op_type == "ConvTranspose":
                need_transpose.append(node.input[1])
    dev = torch.device("cuda", torch.cuda.current_device())
    resident = getattr(session, "consts", None) or {}
    todo = [(name, np.asarray(t)) for name, t in weight_tensor.items() if np.asarray(t).ndim >= 1]
    if not todo:
        return {}
    # every row's (min, max) into ONE device buffer, read back in one transfer (a round trip per initializer — 161 of them for
    # ResNet-50 — was a tenth of a second of a run)
    rows = [t.shape[1] if name in need_transpose else t.shape[0] for name, t in todo]
    res = torch.empty(2, sum(rows), dtype=torch.float32, device=dev)
    off = 0
    for (name, tensor), c in zip(todo, rows):
        w = resident.get(name)
        if w is not None and w.is_cuda and w.dtype == torch.float32 and tuple(w.shape) == tuple(tensor.shape):
            if name in need_transpose:
                w = w.permute(1, 0, 2, 3)
            w2 = w.reshape(c, -1).contiguous()
        else:
            if name in need_transpose:
                tensor = tensor.transpose([1, 0, 2, 3])
            w2 = torch.from_numpy(np.ascontiguousarray(tensor.reshape(c, -1), dtype=np.float32)).to(dev, non_blocking=True)
        ops.rowwise_minmax(w2, out=(res[0, off:off + c], res[1, off:off + c]))
        off += c
    flat = res.cpu().numpy()
    out, off = {}, 0
    for (name, _), c in zip(todo, rows):
        out[name] = [flat[0, off:off + c].copy(), flat[1, off:off + c].copy()]
        off += c
    return out
