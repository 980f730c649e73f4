"""Layer-wise (AdaRound) and block-wise (BRECQ / QDrop) weight-rounding reconstruction on one MI355X per rank —
the machinery behind adaround.py / brecq.py.

Reference (adaround.py:19-116, brecq.py:20-155): for every learnable node it pulls the node's quantised input and
full-precision output for the whole shard out of two host-side ActivationCaches (one ONNXRuntime session per node),
stacks them with numpy, uploads them, learns the round mask with eager torch + DDP, writes the rounded weight back
and lets the caches recompute downstream activations lazily.

Here both networks stay in HBM: the full-precision activations come from one batched forward (ActivationCache), the
fake-quantised network is walked ONCE, node-major (every node is run for the whole shard before the next one, so
"everything downstream sees the rounded weights" costs O(nodes) node executions), and a learning iteration is four
launches around the layer's convolution (ada_quant_layer.py).  Ranks learn the same layer together: dL/d(qw) is
summed over ranks with one RCCL all-reduce per layer and iteration (DDP's mean, adaround.py:121).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from ..executor import GraphSession
from ..forward_net import ActivationCache, load_input_batch
from ..graph import ONNXGraph
from ..platform_settings import platform_setting_table
from ..quantize import quant_graph
from ..utils import logger
from .ada_quant_layer import AdaQLayer, L2_norm, RoundSchedule, adaround_reg
from .bias_correction import _Frontier
from .sparse_quant_layer import SparseQLayer, cosine_lr
from .weight_equalization import node_has_equalized
from .utils import (LEARNABLE_LAYER_TYPES, follow_relu, following_relu, get_block_from_first, get_quant_tensor,
                    update_weight)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _use_graph(world):
    """hipGraph replay of the iteration is opt-in (DPL_ROUND_GRAPH=1): measured on MI355X the loop is bound by the
    two library convolutions, not by launches (0.41 ms eager vs 0.42 ms replayed on ResNet-50 layer shapes), and
    with several ranks the gradient all-reduce is issued from the host between backward and update."""
    return world == 1 and os.environ.get("DPL_ROUND_GRAPH", "0") == "1"


def learn_rounding(layers, q_in, fp_in, fp_out, reg, batch_size, max_epoch, drop=False, log_every=50, log_head="",
                   on_step=None, use_graph=None):
    """adaround.py:119-144 / brecq.py:158-200 — learn the round masks of `layers` (applied in sequence) so that
    layers(q_in) reproduces fp_out.  q_in / fp_in / fp_out: device tensors [n, ...].  Returns the last logged
    (l2, regulariser) pair.  on_step(iteration, layers) is called after every update (tests).

    One iteration = conv forward, fused L2 loss + gradient, conv backward, one fused update per layer.  Its launch
    sequence does not depend on the data and the regulariser temperature / Adam bias corrections are advanced on
    the device (RoundSchedule), so with use_graph every batch index gets its iteration captured as a hipGraph after
    the first (eager, library warm-up) epoch and the remaining epochs are graph replays."""
    world = _world()
    if use_graph is None:
        use_graph = _use_graph(world)
    n = q_in.shape[0]
    n_batches = math.ceil(n / batch_size)
    ratio = 0.5 if drop else 1.0
    last = len(layers) - 1
    fused_relu = layers[last].relu_flag and not layers[last].acti_quant   # the ReLU goes into the loss kernel
    loss = torch.zeros(2, dtype=torch.float64, device=q_in.device)       # [L2, regulariser] of the last iteration
    sched = RoundSchedule(reg.temp_anneal.t_max, q_in.device)
    in_tensor = q_in if ratio >= 1.0 else torch.empty_like(q_in)          # static: the graphs read it in place

    def iteration(idx):
        st = idx * batch_size
        sched.advance()
        z = in_tensor[st:st + batch_size]
        for li, layer in enumerate(layers):
            z = layer(z, apply_relu=not (li == last and fused_relu))
        loss.zero_()
        _, grad = L2_norm(z, fp_out[st:st + batch_size], relu=fused_relu, loss=loss[0:1])
        z.backward(grad)
        for layer in layers:
            if world > 1:
                dist.all_reduce(layer.rp.qw.grad)
            layer.rp.step(0.0, reg.alpha, 1.0 / world, reg_loss=loss[1:2], sched=sched.buf)

    graphs = {}
    cur_iter = 0
    shown = (0.0, 0.0)
    for epoch in range(max_epoch):
        if ratio < 1.0:   # brecq.py:170-173 — a fresh mix of quantised and full-precision block inputs per epoch
            torch.where(torch.rand_like(q_in) < ratio, q_in, fp_in, out=in_tensor)
        for idx in range(n_batches):
            if use_graph and epoch >= 1:
                g = graphs.get(idx)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        iteration(idx)
                    graphs[idx] = g
                g.replay()
            else:
                iteration(idx)
            cur_iter += 1
            if on_step is not None:
                on_step(cur_iter, layers)
        if epoch % log_every == 0:
            l2, rg = (float(v) for v in loss.tolist())
            shown = (l2, rg)
            reg.beta = sched.state()[2]
            if _rank() == 0:
                logger.info("{}Epoch: {:<5} L2 Loss: {:>10.3f} Beta: {:>3.3f}".format(log_head, epoch, l2 + rg, reg.beta))
    reg.beta = sched.state()[2]
    for layer in layers:
        layer.rp.steps = cur_iter
    if _rank() == 0:
        for layer in layers:
            c, f, t = layer.rp.rounding_summary()
            logger.info("Ceil: {:>5} Floor: {:>5} Total: {:>5} Ratio: {:>.3f}".format(c, f, t, (c + f) / t))
    return shown


def learn_sparse(layer, q_in, fp_out, batch_size, max_epoch, log_every=50):
    """sparse_quant.py:107-130 — SGD(lr 1e-3, momentum 0.9, weight decay 1e-4) with a per-epoch cosine schedule on
    the layer's weight; one iteration = mask + fused quantiser, conv forward, fused L2 loss + gradient, conv
    backward, fused straight-through gradient + SGD update."""
    world = _world()
    n = q_in.shape[0]
    n_batches = math.ceil(n / batch_size)
    loss = torch.zeros(1, dtype=torch.float64, device=q_in.device)
    for epoch in range(max_epoch):
        lr = cosine_lr(layer.base_lr, epoch, max_epoch)
        for idx in range(n_batches):
            st = idx * batch_size
            z = layer(q_in[st:st + batch_size], apply_relu=False)
            loss.zero_()
            _, grad = L2_norm(z, fp_out[st:st + batch_size], relu=layer.relu_flag, loss=loss)
            z.backward(grad)
            if world > 1:
                dist.all_reduce(layer.qw.grad)
            layer.step(lr, 1.0 / world)
        if epoch % log_every == 0 and _rank() == 0:
            logger.info("Epoch: {:<4} L2 Loss: {:>10.6f}, LR: {:>10.6f}".format(epoch, float(loss), lr))
    if _rank() == 0:
        logger.info("Loss: {:>10.6f}".format(float(loss)))
    return float(loss)


def _cat(chunks):
    return chunks[0] if len(chunks) == 1 else torch.cat(chunks)


def _build_layer(graph, graph_new, node, clip_val, args, reg, dev, with_acti):
    """adaround.py:50-92 / brecq.py:66-112 — the AdaQLayer of one node from the current weights and ranges."""
    if args.deploy == "nnie":
        raise NotImplementedError("the nnie log-domain rounding is not built")
    plat = platform_setting_table[args.deploy]
    weight = torch.from_numpy(np.ascontiguousarray(graph_new.get_initializer(node.input[1]), dtype=np.float32)).to(dev)
    bias = None
    if len(node.input) == 3:
        bias = torch.from_numpy(np.ascontiguousarray(graph_new.get_initializer(node.input[2]), dtype=np.float32)).to(dev)
    qw_param = plat["qw_params"]
    w_shape = list(weight.shape)
    if node.op_type == "ConvTranspose":
        w_shape[0], w_shape[1] = w_shape[1], w_shape[0]
    scale, q_min, q_max = get_quant_tensor(w_shape, qw_param, clip_val[node.input[1]], dev)
    qw_tensor = {"scale": scale, "q_min": q_min, "q_max": q_max, "per_channel": bool(qw_param.get("per_channel")),
                 "type": "Linear"}
    relu_flag = follow_relu(graph, node)
    qi_tensor = None
    if with_acti:
        out_node = following_relu(graph, node) if relu_flag else node
        a_scale, a_min, a_max = get_quant_tensor(graph.get_tensor_shape(out_node.output[0]), plat["qi_params"],
                                                 clip_val[out_node.output[0]], dev)
        if a_scale.numel() != 1:
            raise NotImplementedError("per-channel activation quantisation inside a block")
        qi_tensor = {"scale": a_scale, "q_min": a_min, "q_max": a_max, "type": "Linear"}
    return AdaQLayer(node, weight, bias, qw_tensor, qi_tensor, relu_flag, with_acti)


def _build_sparse_layer(graph, graph_new, node, clip_val, args, dev):
    """sparse_quant.py:57-84."""
    plat = platform_setting_table[args.deploy]
    weight = torch.from_numpy(np.ascontiguousarray(graph_new.get_initializer(node.input[1]), dtype=np.float32)).to(dev)
    bias = None
    if len(node.input) == 3:
        bias = torch.from_numpy(np.ascontiguousarray(graph_new.get_initializer(node.input[2]), dtype=np.float32)).to(dev)
    w_shape = list(weight.shape)
    if node.op_type == "ConvTranspose":
        w_shape[0], w_shape[1] = w_shape[1], w_shape[0]
    scale, q_min, q_max = get_quant_tensor(w_shape, plat["qw_params"], clip_val[node.input[1]], dev)
    qw_tensor = {"scale": scale, "q_min": q_min, "q_max": q_max, "per_channel": bool(plat["qw_params"].get("per_channel"))}
    sparse_info = {"sparse": True, "rate": args.sparse_rate, "pattern": args.pattern}
    return SparseQLayer(node, weight, bias, qw_tensor, follow_relu(graph, node), sparse_info)


def reconstruct(graph_ori, graph, act_clip_val, weight_clip_val, args, blockwise, save_name, sparse=False):
    """Shared driver of adaround(), brecq() and sparse_quant(): returns the graph with the learned weights."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    clip_val = {k: [np.copy(v[0]), np.copy(v[1])] for k, v in {**act_clip_val, **weight_clip_val}.items()}
    graph_new = ONNXGraph()
    graph_new.copy_from(graph)
    rank, world = _rank(), _world()
    num_per_rank = args.data_num // world
    st, ed = rank * num_per_rank, rank * num_per_rank + num_per_rank
    dev = torch.device("cuda", torch.cuda.current_device())
    skip = getattr(args, "skip_layers", []) or []
    drop = bool(getattr(args, "drop", False)) and blockwise
    with_acti = bool(getattr(args, "acti_quant", False))
    head = "sparse_quant" if sparse else (("Qdrop" if drop else "Brecq") if blockwise else "Adaround")

    with torch.no_grad():
        fp_cache = ActivationCache(graph_ori, args, st, ed)
        graph_q, _ = quant_graph(graph_new, {k: [np.copy(v[0]), np.copy(v[1])] for k, v in clip_val.items()}, args)
        s_q = GraphSession(graph_q, device=dev)
        chunk = int(getattr(args, "calib_batch", 16) or 16)
        bounds = [(i, min(i + chunk, ed)) for i in range(st, ed, chunk)]
        sizes = [j - i for i, j in bounds]
        shapes = {n: graph_ori.get_tensor_shape(n) for n in graph_ori.network_inputs}
        qf = _Frontier(s_q, graph_q)
        for name in graph_ori.network_inputs:
            qf.env[name] = [load_input_batch(args.input_dir, [name], shapes, i, j, dev)[name] for i, j in bounds]
    ori_nodes = {n.name: n for n in graph.graph.node}
    learnable = [n.name for n in graph_ori.graph.node if n.op_type in LEARNABLE_LAYER_TYPES and n.name not in skip]
    already = set()
    for node in graph_q.graph.node:
        if node.name in s_q._folded:
            continue
        if node.name in learnable and node.name not in already:
            block = get_block_from_first(graph, ori_nodes[node.name], args) if blockwise else [ori_nodes[node.name]]
            if getattr(args, "we", False):      # an equalised layer cannot be mimicked (adaround.py:36-37, brecq.py:39-41)
                if not blockwise and node_has_equalized(graph, block[0]):
                    with torch.no_grad():
                        qf.run(node, len(bounds), sizes)
                    continue
                if blockwise and node_has_equalized(graph, block[-1]):
                    block.pop(-1)
                    if not block:           # (the reference would index an empty list here)
                        with torch.no_grad():
                            qf.run(node, len(bounds), sizes)
                        continue
            if rank == 0:
                logger.info("{} for: {}".format(head, " ".join(b.name for b in block)))
            already.update(b.name for b in block)
            epochs = args.ada_epoch * len(block)
            reg = adaround_reg(epochs * math.ceil(num_per_rank / args.ada_bs))
            if sparse:
                layers = [_build_sparse_layer(graph, graph_new, block[0], clip_val, args, dev)]
            else:
                layers = [_build_layer(graph, graph_new, b, clip_val, args, reg, dev, with_acti) for b in block]
            with torch.no_grad():
                q_in = _cat(qf.env[node.input[0]])            # already the fake-quantised tensor if the platform quantises it
                fp_in = _cat(fp_cache.chunks(block[0].input[0])) if drop else None
                fp_out = _cat(fp_cache.chunks(block[-1].output[0]))
                if follow_relu(graph, block[-1]):
                    fp_out = torch.relu(fp_out)
            if sparse:
                learn_sparse(layers[0], q_in, fp_out, args.ada_bs, epochs)
            else:
                learn_rounding(layers, q_in, fp_in, fp_out, reg, args.ada_bs, epochs, drop,
                               log_every=100 if blockwise else 50, log_head="")
            with torch.no_grad():
                for b, layer in zip(block, layers):
                    w_new = layer.new_weight()
                    update_weight(graph_new, w_new.cpu().numpy(), b.input[1])
                    update_weight(graph_q, w_new.cpu().numpy(), b.input[1])
                    s_q.set_const(b.input[1], w_new)
            del layers, q_in, fp_in, fp_out
        with torch.no_grad():
            qf.run(node, len(bounds), sizes)
    graph_new.update_model()
    if rank == 0 and getattr(args, "output_dir", None):
        graph_new.output_dir = args.output_dir
        graph_new.save_onnx_model(save_name)
    return graph_new
