"""Quantisation-accuracy profiling on the GPU — counterpart of dipoorlet/profiling.py:34-99, 246-264.

Reference per image: two fresh ORT sessions (fp model, fake-quantised model), every tensor copied to the
host, three fp32 numpy reductions per compared tensor (utils.py:273-278).  Here: both graphs run batched
on the device, the compared tensors stay in HBM and one `k_cos_items` launch per batch produces
sum(a*b), sum(a*a), sum(b*b) for every (image, tensor) pair in fp64.

Semantics kept: per-layer cosine = mean over this rank's images of the per-image cosine (0.0 when
sum(a*b) == 0); network outputs with <= 10 elements per image ("single") use ONE cosine over the stack
of all images; others report [mean, min].  Sharding is the reference's ceil split (profiling.py:48-51).
Merge over ranks: all-reduce of the sums (SUM) and minima (MIN) instead of per-rank JSON files — for
world_size 1 identical to the reference, for more ranks the exact whole-set mean.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .forward_net import load_input_batch
from .quantize import DQTENSORSUFFIX, quant_graph
from .utils import logger


def get_output_single_map(graph):
    """profiling.py:200-207 — outputs with <= 10 values per image are compared as one stacked vector."""
    return {o: int(np.prod(graph.get_tensor_shape(o)[1:])) <= 10 for o in graph.network_outputs}


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def quantize_profiling_multipass(graph_after_wt, graph_ori, act_clip_val, weight_clip_val, args):
    """profiling.py:34-99 -> (layer_cosine_dict, model_cosine_dict, quant_node_list)."""
    clip_val = act_clip_val.copy()
    clip_val.update(weight_clip_val)
    graph_q, quant_node_list = quant_graph(graph_after_wt, clip_val, args)
    rank = dist.get_rank() if _world() > 1 or (dist.is_available() and dist.is_initialized()) else 0
    if rank == 0 and getattr(args, "output_dir", None):
        graph_q.output_dir = args.output_dir
        graph_q.save_onnx_model(name="quant_model")
    dev = torch.device("cuda", torch.cuda.current_device())
    fp_sess = graph_ori.make_session(args)
    q_sess = graph_q.make_session(args)
    world = getattr(args, "world_size", 1)
    per = math.ceil(args.data_num / world)
    st, ed = getattr(args, "rank", 0) * per, min(getattr(args, "rank", 0) * per + per, args.data_num)
    layer_names = [t for node in quant_node_list for t in node.output]
    outs = list(graph_after_wt.network_outputs)
    single = get_output_single_map(graph_after_wt)
    q_out_name = {o: (o + DQTENSORSUFFIX if (o + DQTENSORSUFFIX) in graph_q.output_map else o) for o in outs}
    names_fp = layer_names + outs
    names_q = layer_names + [q_out_name[o] for o in outs]
    shapes = {n: graph_ori.get_tensor_shape(n) for n in graph_ori.network_inputs}
    batch = int(getattr(args, "calib_batch", 16) or 16)
    plans = {}
    layer_sum = torch.zeros(len(layer_names), dtype=torch.float64, device=dev)
    out_sum = torch.zeros(len(outs), dtype=torch.float64, device=dev)
    out_min = torch.full((len(outs),), float("inf"), dtype=torch.float64, device=dev)
    out_tot = torch.zeros(len(outs), 3, dtype=torch.float64, device=dev)   # stacked-vector sums for "single"
    n_img = 0
    i = st
    while i < ed:
        j = min(i + batch, ed)
        b = j - i
        inputs = load_input_batch(args.input_dir, graph_ori.network_inputs, shapes, i, j, dev)
        fp = [t.float().contiguous() for t in fp_sess.run_named(inputs, names_fp)]
        qq = [t.float().contiguous() for t in q_sess.run_named(inputs, names_q)]
        plan = plans.get(b)
        if plan is None:
            plan = plans[b] = ops.TensorSetPlan([t.numel() // b for t in fp], b, dev)
        sums = ops.cos_per_image(plan, fp, qq)                      # [b, T, 3]
        ab, aa, bb = sums[..., 0], sums[..., 1], sums[..., 2]
        cos = torch.where(ab == 0, torch.zeros_like(ab), ab / torch.sqrt(aa) / torch.sqrt(bb))
        L = len(layer_names)
        layer_sum += cos[:, :L].sum(0)
        out_sum += cos[:, L:].sum(0)
        out_min = torch.minimum(out_min, cos[:, L:].min(0).values)
        out_tot += sums[:, L:, :].sum(0)
        if getattr(args, "savefp", False) and rank == 0:
            import os
            for k, o in enumerate(outs):
                d = os.path.join(args.output_dir, "output", o)
                os.makedirs(d, exist_ok=True)
                for r in range(b):
                    fp[L + k][r].cpu().numpy().astype(np.float32).tofile(os.path.join(d, f"onnx-output-{i + r}.bin"))
        n_img += b
        i = j
    cnt = torch.tensor([float(n_img)], dtype=torch.float64, device=dev)
    if world > 1 and dist.is_initialized():
        for t in (layer_sum, out_sum, out_tot, cnt):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(out_min, op=dist.ReduceOp.MIN)
    n = float(cnt.item())
    layer_cosine = {t: float(v) / n for t, v in zip(layer_names, layer_sum.cpu().tolist())}
    model_cosine = {}
    tot = out_tot.cpu().numpy()
    for k, o in enumerate(outs):
        if single[o]:
            ab, aa, bb = tot[k]
            c = 0.0 if ab == 0 else float(ab / np.sqrt(aa) / np.sqrt(bb))
            model_cosine[o] = [c, c]
        else:
            model_cosine[o] = [float(out_sum[k].item()) / n, float(out_min[k].item())]
    return layer_cosine, model_cosine, quant_node_list


def quantize_profiling_transformer(graph_after_wt, graph_ori, act_clip_val, weight_clip_val, args):
    """profiling.py:102-156 — the `--model_type` / `--optim_transformer` variant: network-output cosines only (its
    layer dictionary stays empty).  The reference needs a separate node-by-node executor for large transformer
    graphs; here the same batched pass serves both, so this is the multipass result without the per-layer part."""
    _, model_cosine, quant_node_list = quantize_profiling_multipass(graph_after_wt, graph_ori, act_clip_val,
                                                                    weight_clip_val, args)
    return {}, model_cosine, quant_node_list


def show_model_profiling_res(graph_after_wt, layer_cosine_dict, model_cosine_dict, quant_node_list, args):
    """profiling.py:246-264 — log per-layer cosines, the 10 worst layers, and the output cosines."""
    import heapq
    single = get_output_single_map(graph_after_wt)
    heap = []
    if not getattr(args, "skip_prof_layer", False):
        for node in quant_node_list:
            logger.info(node.name)
            for t in node.output:
                logger.info("Layer  cos: {:.5f}".format(layer_cosine_dict[t]))
                heapq.heappush(heap, (layer_cosine_dict[t], node.name + "-" + t))
        logger.info("The smallest cos value of 10 layers: ")
        for c, name in heapq.nsmallest(10, heap):
            logger.info("{:40} cos : {:<.5f}".format(name, c))
    logger.info("Quant model output cos: ")
    for name in graph_after_wt.network_outputs:
        if not single[name]:
            logger.info("{:40} avgcos : {:<.5f}    mincos : {:<.5f}".format(name, *model_cosine_dict[name]))
        else:
            logger.info("{:40} tolcos : {:<.5f}".format(name, model_cosine_dict[name][0]))


def show_model_ranges(graph, act_clip_val, weight_clip_val, args):
    """profiling.py:210-224 — log every activation / weight range with its tensor shape."""
    from .platform_settings import platform_setting_table
    logger.info("Model ranges:")
    ranges_all = act_clip_val.copy()
    ranges_all.update(weight_clip_val)
    per_channel = "per channel " if platform_setting_table[args.deploy]["qw_params"].get("per_channel", False) else ""
    for name, rng in ranges_all.items():
        shape = str(graph.tensor_name_shape_map.get(name))
        if isinstance(rng[0], np.ndarray) and rng[0].ndim > 0:
            logger.info("{:<30} Shape: {:<20} Range: {}[{:<10f} {:<10f}]".format(name, shape, per_channel,
                                                                                 rng[0].min(), rng[1].max()))
        else:
            logger.info("{:<30} Shape: {:<20} Range: [{:<10f} {:<10f}]".format(name, shape, float(rng[0]), float(rng[1])))


def weight_need_perchannel(graph, args):
    """profiling.py:227-243 — for per-tensor weight platforms, rank Conv layers by mean per-channel range /
    per-layer range (a small ratio means per-layer quantisation wastes most of the grid on that layer)."""
    import heapq

    from .platform_settings import platform_setting_table
    if platform_setting_table[args.deploy]["qw_params"].get("per_channel", False):
        return
    logger.info("Layer degradate by per layer: ")
    heap = []
    for node in graph.graph.node:
        if node.op_type == "Conv":
            w = np.asarray(graph.get_initializer(node.input[1]))
            w2 = w.reshape(w.shape[0], -1)
            ratio = (w2.max(-1) - w2.min(-1)).mean() / (w.max() - w.min())
            heapq.heappush(heap, (float(ratio), node.name))
    for ratio, name in heapq.nsmallest(len(heap), heap):
        logger.info("{:40} ratio : {:<.5f}".format(name, ratio))
