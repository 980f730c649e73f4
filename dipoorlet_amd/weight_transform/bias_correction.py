"""`--bc` bias correction on the GPU — counterpart of dipoorlet/weight_transform/bias_correction.py:9-55
(and its call site weight_trans_base.py:21-29).

Reference: for every Conv / Gemm node in topological order it re-builds the fake-quantised graph,
creates one ONNXRuntime session per node through the host-side ActivationCache (forward_net.py:23-190),
pulls the node's fp and quantised outputs for ALL N images to the host and adds mean(fp - q) over
(N, H, W) to the bias, so that later nodes see the corrected upstream biases.

Here the whole calibration set's frontier stays resident in HBM and the two networks (fp and
fake-quantised) are walked ONCE, node-major: each node is executed for all images before the next one
starts (activations are kept as lists of per-chunk device tensors and freed by reference count), so the
sequential dependence "correct bias k, then everything downstream sees it" costs O(nodes) node
executions instead of O(nodes^2).  The bias enters a Conv / Gemm output linearly, so after the update the
already computed quantised output is fixed up in place (q_out += diff) instead of being recomputed.
The fake-quant structure does not depend on bias values, so the quantised graph is built once.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..executor import _OPS, GraphSession, fused_fake_quant, relu_fusion
from ..forward_net import load_input_batch
from ..graph import ONNXGraph
from ..quantize import quant_graph
from ..utils import logger

BIAS_CORRECTION_NODE_TYPE = ["Conv", "Gemm"]


def _channel_mean_diff(fp_chunks, q_chunks, is_conv, world_size=1, n_ch=None):
    """bias_correction.py:10-13 — mean(fp - q) over every axis but the channel one (axis 1 of a Conv output
    [n, C, spatial...]; the last axis of a Gemm output [n, C]): one fused kernel per chunk pair, fp64 sums.
    world_size > 1: the chunks are this rank's shard of the images; the per-channel fp64 sums and the count are added up over
    the ranks (ONE all-reduce of [C + 1] doubles per node), so every rank returns the mean over the whole set."""
    acc, cnt = None, 0
    dev = torch.device("cuda", torch.cuda.current_device())
    if n_ch is not None:        # (a rank whose shard is empty still joins the all-reduce)
        acc = torch.zeros(int(n_ch), dtype=torch.float64, device=dev)
    for a, b in zip(fp_chunks, q_chunks):
        a, b = a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)   # (no-ops unless the frontier is kept on the host)
        if not is_conv:
            a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
        acc = ops.channel_diff_sum(a.contiguous(), b.contiguous(), acc)
        cnt += a.numel() // a.shape[1]
    if world_size > 1:
        packed = torch.cat([acc, torch.tensor([float(cnt)], dtype=torch.float64, device=acc.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return (packed[:-1] / packed[-1]).float()
    return (acc / cnt).float()


def update_conv_node_bias(graph_bc, node, fp_activations, q_activations, world_size=1):
    """bias_correction.py:9-31 on device tensors: bias += mean(fp - q) over every axis but the channel one (Conv output
    chunks [n, C, H, W], Gemm output chunks [n, C]); a node without a bias input gets `<node>_bias`.  Returns the
    per-channel difference (fp32 device tensor) that was added.  `*_activations`: lists of per-chunk device tensors
    (world_size > 1: of this rank's shard; the mean is over all ranks' images)."""
    bc_node = next(n for n in graph_bc.graph.node if n.name == node.name)
    n_ch = None
    if world_size > 1:          # the channel count from the weights: [C_out, ...] (Conv; Gemm with transB), [K, C_out] (Gemm)
        w = graph_bc.get_initializer(bc_node.input[1])
        n_ch = w.shape[0] if (node.op_type == "Conv" or bc_node.attrs.get("transB", 0)) else w.shape[1]
    diff = _channel_mean_diff(fp_activations, q_activations, node.op_type == "Conv", world_size, n_ch)
    if len(bc_node.input) > 2:
        bname = bc_node.input[2]
        new_bias = graph_bc.get_initializer(bname).astype(np.float32) + diff.cpu().numpy()
    else:
        bname = node.name + "_bias"
        new_bias = diff.cpu().numpy()
        bc_node.input.append(bname)
        graph_bc.input.append(bname)
    graph_bc.set_initializer(bname, new_bias.astype(np.float32))
    return diff


def bc_shard(data_num, rank, world_size):
    """Images [st, ed) of rank `rank` for the sharded --bc walk: a balanced split that covers ALL data_num images (the reference
    corrects with every image, forward_net.py:50-52; the calibration sweeps' floor split would drop data_num % world_size)."""
    return rank * data_num // world_size, (rank + 1) * data_num // world_size


def _frontier_peak_elems(graph, session):
    """Largest number of live activation elements per image during a node-major walk (reference-count simulation over
    the per-image tensor sizes the session inferred)."""
    size = dict(zip(session.tensor_names, session.elems_per_image))
    ref = {}
    for node in graph.graph.node:
        if node.name in session._folded:
            continue
        for i in node.input:
            if i != "" and i not in session.consts:
                ref[i] = ref.get(i, 0) + 1
    for o in graph.network_outputs:
        ref[o] = ref.get(o, 0) + 1
    live = {n: size.get(n, 0) for n in graph.network_inputs}
    peak = sum(live.values())
    for node in graph.graph.node:
        if node.name in session._folded:
            continue
        for o in node.output:
            if o != "":
                live[o] = size.get(o, size.get(node.input[0], 0) if node.input else 0)
        peak = max(peak, sum(live.values()))
        for i in node.input:
            if i in ref:
                ref[i] -= 1
                if ref[i] == 0:
                    live.pop(i, None)
        for o in node.output:
            if o not in ref:
                live.pop(o, None)
    return peak


class _Frontier:
    """Activations of every live tensor for the whole calibration set, as lists of per-chunk tensors."""

    def __init__(self, session, graph, on_host=False, keep=()):
        """keep: tensors the caller reads from `env` by name besides a node's own inputs / outputs while it runs.  On a
        fake-quantised graph a ReLU (and the residual Add in front of it) whose only reader is a Q/DQ pair runs inside that pair's
        kernel (executor.relu_fusion): its output is never in `env`."""
        self.sess, self.graph = session, graph
        self.env = {}
        self.ref = {}
        self.fused, self.skipped = relu_fusion(graph, session._folded, session.consts, keep, getattr(session, "shape1", None))
        # on_host: the live activations of the whole set do not fit the HBM budget — chunks wait in (pinned) host memory and
        # come back to the device one at a time when a node consumes them: slower (PCIe both ways), same values
        self.on_host = on_host
        self.dev = torch.device("cuda", torch.cuda.current_device())
        for node in graph.graph.node:
            if node.name in session._folded or node.name in self.skipped:
                continue
            for i in self.inputs_of(node):
                if i != "" and i not in session.consts:
                    self.ref[i] = self.ref.get(i, 0) + 1
        for o in graph.network_outputs:
            self.ref[o] = self.ref.get(o, 0) + 1

    def inputs_of(self, node):
        """The tensors `node` reads: its inputs, or — a Q/DQ pair that runs its producers' Add / ReLU — theirs."""
        return self.fused[node.name][1] if node.name in self.fused else node.input

    def run(self, node, n_chunks, chunk_sizes):
        if node.name in self.skipped:       # runs inside the Q/DQ kernel behind it
            return
        ins = self.inputs_of(node)
        outs = [[] for _ in node.output]
        for c in range(n_chunks):
            self.sess.batch = chunk_sizes[c]
            args = [None if i == "" else (self.sess.consts[i] if i in self.sess.consts else self.env[i][c].to(self.dev))
                    for i in ins]
            while args and args[-1] is None:
                args.pop()
            if node.name in self.fused:
                r = fused_fake_quant(self.sess, node, self.fused[node.name][0], *args)
            else:
                r = _OPS[node.op_type](self.sess, node, *args)
            r = list(r) if isinstance(r, (list, tuple)) else [r]
            for k, v in enumerate(r[:len(outs)]):
                outs[k].append(v.cpu() if self.on_host else v)
        for o, v in zip(node.output, outs):
            if o != "":
                self.env[o] = v
        for i in ins:
            if i in self.ref:
                self.ref[i] -= 1
                if self.ref[i] == 0:
                    self.env.pop(i, None)
        for o in node.output:  # outputs nobody consumes
            if o not in self.ref:
                self.env.pop(o, None)


@torch.no_grad()
def bias_correction(graph, act_clip_val, weight_clip_val, args):
    """bias_correction.py:34-55 -> the bias-corrected graph (also saved as update_bias_model.onnx)."""
    clip_val = act_clip_val.copy()
    clip_val.update(weight_clip_val)
    clip_val = {k: [np.copy(v[0]), np.copy(v[1])] for k, v in clip_val.items()}
    graph_bc = ONNXGraph()
    graph_bc.copy_from(graph)
    graph_q, _ = quant_graph(graph_bc, clip_val, args)
    dev = torch.device("cuda", torch.cuda.current_device())
    s_fp = GraphSession(graph, device=dev)
    s_q = GraphSession(graph_q, device=dev)
    chunk = int(getattr(args, "calib_batch", 16) or 16)
    # The reference lets rank 0 walk ALL images while the others wait (weight_trans_base.py:21-29, forward_net.py:50-52).  The
    # correction is a per-channel SUM over images, so with several ranks each walks its shard of the images node-major as
    # before and the sums are all-reduced per Conv / Gemm node (RCCL): every rank holds the same corrected biases, and its
    # frontier is 1 / world of the set.  args.merge == 'reference' keeps the reference's schedule.
    world = int(getattr(args, "world_size", 1) or 1)
    sharded = world > 1 and getattr(args, "merge", "allreduce") != "reference" and dist.is_available() and dist.is_initialized()
    st, ed = bc_shard(args.data_num, int(getattr(args, "rank", 0)), world) if sharded else (0, args.data_num)
    world = world if sharded else 1
    N = ed - st
    bounds = [(i, min(i + chunk, ed)) for i in range(st, ed, chunk)]
    sizes = [j - i for i, j in bounds]
    shapes = {n: graph.get_tensor_shape(n) for n in graph.network_inputs}
    # HBM budget: the two frontiers hold the WHOLE set's live activations.  Refuse up front with a plain message rather than
    # die in the allocator halfway through (the other ranks would be left at the next barrier).
    need = 4.0 * N * (_frontier_peak_elems(graph, s_fp) + _frontier_peak_elems(graph_q, s_q))
    budget = float(getattr(args, "resident_gb", 160.0) or 160.0) * 1e9
    on_host = need > budget
    if on_host:
        logger.warning("--bc: the live activations of all %d images of both networks are about %.0f GB at the widest point of "
                       "this graph, over the %.0f GB budget (--resident_gb): keeping them in host memory between nodes",
                       N, need / 1e9, budget / 1e9)
    fp, qf = _Frontier(s_fp, graph, on_host), _Frontier(s_q, graph_q, on_host)
    for n in graph.network_inputs:
        chunks = [load_input_batch(args.input_dir, [n], shapes, i, j, dev)[n] for i, j in bounds]
        if on_host:
            chunks = [t.cpu() for t in chunks]
        fp.env[n] = chunks
        qf.env[n] = chunks
    fp_nodes = {n.name: n for n in graph.graph.node}
    # DPL_BC_RECOMPUTE=1 (a testing aid): a corrected node's quantised output is computed AGAIN with the corrected bias instead of
    # being fixed up in place (q_out + diff).  The two are the same value up to one fp32 rounding — conv(x, w, b) + d against
    # conv(x, w, b + d) — but a last-bit difference flips a rounding step of a fake-quantised layer downstream now and then; with
    # the recomputation the walk IS the reference's definition evaluated node-major (tests/test_cli_e2e.py compares them exactly).
    recompute = os.environ.get("DPL_BC_RECOMPUTE") == "1"
    for node in graph_q.graph.node:
        if node.name in s_q._folded:
            continue
        if node.name not in fp_nodes:
            qf.run(node, len(bounds), sizes)
            continue  # a FakeQuant node
        fp_node = fp_nodes[node.name]
        corrected = node.op_type in BIAS_CORRECTION_NODE_TYPE
        out = node.output[0]
        if corrected:
            fp.ref[fp_node.output[0]] = fp.ref.get(fp_node.output[0], 0) + 1  # hold both outputs for the diff BEFORE the nodes
            qf.ref[out] = qf.ref.get(out, 0) + 1                              # run: an output nobody consumes is dropped at once
            if recompute:                                                     # (... and the inputs for the second run)
                for i in qf.inputs_of(node):
                    if i in qf.ref:
                        qf.ref[i] += 1
        qf.run(node, len(bounds), sizes)
        fp.run(fp_node, len(bounds), sizes)
        if not corrected:
            continue
        logger.info("Update bias for node: {}".format(node.name))
        bc_node = next(n for n in graph_bc.graph.node if n.name == node.name)
        diff = update_conv_node_bias(graph_bc, node, fp.env[out], qf.env[out], world)
        if recompute:
            bname = bc_node.input[2]
            if len(node.input) < 3:       # (a node without a bias has just been given one)
                node.input.append(bname)
            s_q.set_const(bname, torch.from_numpy(np.ascontiguousarray(graph_bc.get_initializer(bname), dtype=np.float32)))
            qf.run(node, len(bounds), sizes)
        elif qf.env[out]:
            shape = [1, -1] + [1] * (qf.env[out][0].dim() - 2)
            d_host = diff.reshape(shape).cpu() if on_host else None
            for t in qf.env[out]:  # the bias is additive in the output: fix the computed q output in place
                t.add_(d_host if on_host else diff.reshape(shape))
        for env_ref, o in ((fp, fp_node.output[0]), (qf, out)):   # drop the extra hold
            env_ref.ref[o] -= 1
            if env_ref.ref[o] == 0:
                env_ref.env.pop(o, None)
    graph_bc.update_model()
    if getattr(args, "output_dir", None) and (not sharded or int(getattr(args, "rank", 0)) == 0):
        graph_bc.output_dir = args.output_dir
        graph_bc.save_onnx_model("update_bias_model")
    return graph_bc
