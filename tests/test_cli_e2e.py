"""GPU: the CLI end to end from a real .onnx file and a .bin calibration directory (BASELINE configs[0]
shape: ResNet-18, minmax / hist / mse), checked against the CPU oracle applied to the very activations
the executor produced."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

N, BATCH, IMG = 8, 4, 64


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from dipoorlet_amd import models
    d = tmp_path_factory.mktemp("cli")
    g = models.resnet18(seed=11, image=IMG)
    g.output_dir = str(d)
    g.save_onnx_model("model")
    os.makedirs(d / "calib" / "input")
    rng = np.random.default_rng(5)
    for i in range(N):
        rng.standard_normal(3 * IMG * IMG).astype(np.float32).tofile(d / "calib" / "input" / f"{i}.bin")
    return d


@pytest.fixture(scope="module")
def activations(workdir):
    """All calibration tensors of all images, produced with the same batch composition the CLI uses."""
    from dipoorlet_amd.forward_net import load_input_batch
    from dipoorlet_amd.graph import ONNXGraph
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    s = g.make_session()
    acts = {n: [] for n in s.tensor_names}
    for i in range(0, N, BATCH):
        inp = load_input_batch(str(workdir / "calib"), g.network_inputs, {"input": g.get_tensor_shape("input")}, i,
                               i + BATCH, torch.device("cuda:0"))
        for n, t in zip(s.tensor_names, s.run(inp)):
            acts[n] += [x.reshape(-1) for x in t.cpu().numpy()]
    return g, acts


def _run(workdir, algo, deploy, extra=()):
    """Runs the CLI in-process and records, on the host, exactly the activations its calibration forward
    produced (convolution algorithms are not bit-reproducible from one session to the next, and a
    percentile clip moves by a whole bin if a value crosses an edge)."""
    from dipoorlet_amd.__main__ import main
    from dipoorlet_amd.executor import GraphSession
    out = workdir / f"out_{algo}_{deploy}"
    rec = {}
    orig = GraphSession.run

    def spy(self, inputs):
        res = orig(self, inputs)
        for n, t in zip(self.tensor_names, res):
            rec.setdefault(n, []).extend(x.reshape(-1) for x in t.cpu().numpy())
        return res
    GraphSession.run = spy
    try:
        rc = main(["-M", str(workdir / "model.onnx"), "-I", str(workdir / "calib"), "-N", str(N), "-A", algo, "-D",
                   deploy, "-O", str(out), "--calib_batch", str(BATCH), *extra])
    finally:
        GraphSession.run = orig
    assert rc == 0
    assert all(len(v) == N for v in rec.values()), "one calibration forward per image expected"
    return out, rec


def test_cli_hist_trt(workdir, activations):
    g, _ = activations
    out, acts = _run(workdir, "hist", "trt")
    act = json.load(open(out / "act_clip_val.json"))
    wt = json.load(open(out / "weight_clip_val.json"))
    trt = json.load(open(out / "trt_clip_val.json"))["blob_range"]
    assert set(act) == set(acts) and len(act) == 50
    for name, per_img in acts.items():
        lo = min(O.minmax(x)[0] for x in per_img)
        hi = max(O.minmax(x)[1] for x in per_img)
        h = sum(O.abs_hist(x, 2048, O.hist_dmax(lo, hi)) for x in per_img)
        clip = O.hist_percentile(h, lo, hi, 2048, 0.99999)
        assert act[name] == [float(clip[0]), float(clip[1])], name
        assert trt[name] == max(-float(clip[0]), float(clip[1]))
    for name, arr in g.initializer.items():   # per-output-channel weight ranges (every Conv/Gemm initializer)
        a2 = arr.reshape(arr.shape[0], -1)
        assert wt[name][0] == a2.min(-1).tolist() and wt[name][1] == a2.max(-1).tolist()
    assert os.path.getsize(out / "quant_model.onnx") > 1e6  # profiling ran and saved the Q/DQ model
    from dipoorlet_amd import onnx_io
    qm = onnx_io.load_model(str(out / "quant_model.onnx"))
    ops_ = [n.op_type for n in qm.nodes]
    assert ops_.count("QuantizeLinear") == ops_.count("DequantizeLinear") > 20
    assert "conv1.weight_scale" in qm.initializers and qm.initializers["conv1.weight_zero_point"].dtype == np.int8


def test_cli_minmax_and_mse(workdir, activations):
    g, _ = activations
    out, acts = _run(workdir, "minmax", "snpe", ["--skip_profiling"])
    act = json.load(open(out / "act_clip_val.json"))
    enc = json.load(open(out / "snpe_encodings.json"))["activation_encodings"]
    for name, per_img in acts.items():
        lo = min(O.minmax(x)[0] for x in per_img)
        hi = max(O.minmax(x)[1] for x in per_img)
        assert act[name] == [float(lo), float(hi)], name
    assert enc["input"][0]["bitwidth"] == 8 and "output" in enc
    out, acts = _run(workdir, "mse", "ti", ["--skip_profiling"])
    act = json.load(open(out / "act_clip_val.json"))
    for name, per_img in acts.items():
        mins = [O.minmax(x)[0] for x in per_img]
        maxs = [O.minmax(x)[1] for x in per_img]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = [O.octav_scale(x, O.octav_unsigned(mn, True)) for x, mn in zip(per_img, mins)]
            clip = O.octav_clip(s, mins, maxs)
        assert np.allclose(act[name], [float(clip[0]), float(clip[1])], rtol=1e-5, atol=1e-5), (name, act[name], clip)


@pytest.mark.two_forwards
def test_profiling_cosine_against_numpy(workdir, activations):
    import types

    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.profiling import quantize_profiling_multipass
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import tensor_calibration
    from dipoorlet_amd.utils import load_clip_val, save_clip_val
    dist_helper.init_default()
    g, acts = activations
    out = workdir / "prof"
    os.makedirs(out, exist_ok=True)
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, rank=0, local_rank=0, world_size=1,
                                 bins=2048, threshold=0.99999, deploy="trt", act_quant="minmax", calib_batch=BATCH,
                                 output_dir=str(out), skip_layers=[], savefp=False)
    a, w = tensor_calibration(g, args)
    save_clip_val(a, w, args)
    a, w = load_clip_val(args)
    layer, model, qnodes = quantize_profiling_multipass(g, g, a, w, args)
    assert len(layer) == sum(len(n.output) for n in qnodes)
    assert all(0.9 < v <= 1.0 + 1e-9 for v in layer.values()), min(layer.values())
    # independent check of two layers + the output: run the quantised graph, cosine with numpy
    clip = dict(a)
    clip.update(w)
    gq, _ = quant_graph(g, clip, args)
    sq = gq.make_session()
    from dipoorlet_amd.forward_net import load_input_batch
    names = ["conv1_out", "layer2.0.add_out", "output"]
    cos = {n: [] for n in names}
    for i in range(0, N, BATCH):
        inp = load_input_batch(args.input_dir, g.network_inputs, {"input": g.get_tensor_shape("input")}, i, i + BATCH,
                               torch.device("cuda:0"))
        for n, t in zip(names, sq.run_named(inp, names)):
            for r, x in enumerate(t.cpu().numpy()):
                cos[n].append(float(O.cos_similarity(acts[n][i + r], x.reshape(-1))))
    # (the activations are regenerated here — bit for bit, the library runs its deterministic algorithms in this test — and the
    # oracle's cosine is three fp32 numpy reductions where the product sums in fp64: fp32 rounding of sums over <= 16 K elements)
    for n in names[:2]:
        assert abs(layer[n] - np.mean(cos[n])) < 2e-5, (n, layer[n], np.mean(cos[n]))
    assert abs(model["output"][0] - np.mean(cos["output"])) < 2e-5 and abs(model["output"][1] - np.min(cos["output"])) < 2e-5


@pytest.mark.two_forwards
def test_bias_correction_over_hbm_budget_keeps_the_frontier_on_the_host(workdir):
    """--bc beyond the HBM budget (--resident_gb): the whole-set activations wait in host memory between nodes instead of
    the run being refused — slower, and the same corrected biases, bit for bit (deterministic library algorithms: the two walks
    execute the same kernels on the same values; the per-channel sums are fp64)."""
    import types

    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.tensor_cali import tensor_calibration
    from dipoorlet_amd.weight_transform import bias_correction
    dist_helper.init_default()
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    out = workdir / "bc_host"
    os.makedirs(out, exist_ok=True)
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, rank=0, local_rank=0, world_size=1,
                                 bins=2048, threshold=0.99999, deploy="trt", act_quant="minmax", calib_batch=BATCH,
                                 output_dir=str(out), skip_layers=[])
    a, w = tensor_calibration(g, args)
    g_dev = bias_correction(g, a, w, args)
    args.resident_gb = 1e-6                      # nothing fits: every chunk goes to the host and comes back
    g_host = bias_correction(g, a, w, args)
    n_checked = 0
    for node in g.graph.node:
        if node.op_type in ("Conv", "Gemm"):
            bname = next(n for n in g_dev.graph.node if n.name == node.name).input[2]
            assert np.array_equal(g_dev.get_initializer(bname), g_host.get_initializer(bname)), node.name
            n_checked += 1
    assert n_checked >= 10


def _ulps(a, b):
    """|a - b| in units of the fp32 spacing at b."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.maximum(np.abs(b), np.float32(1e-30))).astype(np.float64)


@pytest.mark.two_forwards
def test_bias_correction_matches_sequential_definition(workdir, monkeypatch):
    """--bc: the node-major HBM-resident walk equals the reference's definition evaluated the slow way (for every Conv/Gemm in
    order: fake-quantise the current graph, run BOTH graphs in full over all images, bias += mean(fp - q) over (N, H, W);
    bias_correction.py:34-55) — EXACTLY: the library runs its deterministic algorithms, both sides execute the same batch
    shapes, and with DPL_BC_RECOMPUTE=1 the walk recomputes a corrected node's quantised output (as the definition does) instead
    of fixing it up in place.  What is left between the two is the order of an fp64 sum: at most the last bit of a bias.

    The product's default — the in-place fix-up q_out + diff — is the same value up to one fp32 rounding per element (checked here
    on the first convolution: conv(x, w, b) + d against conv(x, w, b + d)); downstream a last-bit difference flips a rounding
    step of a fake-quantised layer now and then, so its biases are compared with the recomputed ones coarsely (a broken fix-up
    — wrong axis, wrong sign, not applied — moves every later bias by the size of the corrections themselves)."""
    import types

    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.forward_net import load_input_batch
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import tensor_calibration
    from dipoorlet_amd.utils import load_clip_val, save_clip_val
    from dipoorlet_amd.weight_transform import bias_correction
    dist_helper.init_default()
    dev = torch.device("cuda:0")
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    out = workdir / "bc"
    os.makedirs(out, exist_ok=True)
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, rank=0, local_rank=0, world_size=1,
                                 bins=2048, threshold=0.99999, deploy="trt", act_quant="minmax", calib_batch=BATCH,
                                 output_dir=str(out), skip_layers=[])
    a, w = tensor_calibration(g, args)
    save_clip_val(a, w, args)
    a, w = load_clip_val(args)
    g_fix = bias_correction(g, a, w, args)                     # the product's default: fix-up in place
    monkeypatch.setenv("DPL_BC_RECOMPUTE", "1")
    g_bc = bias_correction(g, a, w, args)
    monkeypatch.delenv("DPL_BC_RECOMPUTE")
    assert os.path.exists(out / "update_bias_model.onnx")
    # the slow sequential definition, every Conv / Gemm node
    inp = load_input_batch(args.input_dir, g.network_inputs, {"input": g.get_tensor_shape("input")}, 0, N, dev)
    ref = ONNXGraph()
    ref.copy_from(g)
    s_fp = g.make_session()
    targets = [n for n in g.graph.node if n.op_type in ("Conv", "Gemm")]
    assert len(targets) >= 20
    worst, step = 0.0, 0.0
    for node in targets:
        clip = {k: [np.copy(v[0]), np.copy(v[1])] for k, v in {**a, **w}.items()}
        gq, _ = quant_graph(ref, clip, args)

        def chunked(sess, name):        # (in chunks of BATCH images, as bias_correction walks the set: same shapes, same kernels)
            return torch.cat([sess.run_named({k: v[i:i + BATCH] for k, v in inp.items()}, [name])[0] for i in range(0, N, BATCH)]).double()
        fp_o = chunked(s_fp, node.output[0])
        q_o = chunked(gq.make_session(), node.output[0])
        d = (fp_o - q_o)
        diff = d.mean(dim=(0, 2, 3)) if node.op_type == "Conv" else d.mean(0)
        bname = node.input[2]
        want = (ref.get_initializer(bname) + diff.float().cpu().numpy()).astype(np.float32)
        got = g_bc.get_initializer(bname)
        assert _ulps(got, want).max() <= 1.0, (node.name, float(np.abs(got - want).max()), float(_ulps(got, want).max()))
        assert np.abs(got - g.get_initializer(bname)).max() > 0  # something was corrected
        step = max(step, float(np.abs(got - g.get_initializer(bname)).max()))
        worst = max(worst, float(np.abs(g_fix.get_initializer(bname) - got).max()))
        ref.set_initializer(bname, got.astype(np.float32))
    assert worst <= 0.25 * step, (worst, step)
    # the fix-up itself, where it happens: one convolution, its output with the correction added afterwards against the
    # convolution run with the corrected bias — one more fp32 rounding (and the kernel's own order of adding the bias)
    first = targets[0]
    x = inp["input"][:BATCH]
    wt = torch.from_numpy(g.get_initializer(first.input[1])).to(dev)
    b0 = torch.from_numpy(g.get_initializer(first.input[2])).to(dev)
    dd = torch.from_numpy(g_bc.get_initializer(first.input[2])).to(dev) - b0
    from dipoorlet_amd.executor import _OPS
    sess = g.make_session()
    y_fix = _OPS["Conv"](sess, first, x, wt, b0) + dd.reshape(1, -1, 1, 1)
    y_re = _OPS["Conv"](sess, first, x, wt, b0 + dd)
    eps = float(np.finfo(np.float32).eps)
    assert float((y_fix - y_re).abs().max()) <= 4 * eps * float(y_re.abs().max() + dd.abs().max() + b0.abs().max())


def test_vit_calibration_mse_and_cli_bc(tmp_path):
    """BASELINE configs[4] in miniature: a ViT (decomposed LayerNorm / attention / erf-GELU), -A mse, --bc.
    Per-image OCTAV needs batch-major tensors although the graph carries the batch on axis 1 in places."""
    import types

    from dipoorlet_amd import models
    from dipoorlet_amd.__main__ import main
    from dipoorlet_amd.forward_net import forward_net_octav
    g = models.vit(seed=2, depth=2, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10)
    g.output_dir = str(tmp_path)
    g.save_onnx_model("vit")
    os.makedirs(tmp_path / "calib" / "input")
    rng = np.random.default_rng(9)
    imgs = [rng.standard_normal(3 * 32 * 32).astype(np.float32) for _ in range(6)]
    for i, x in enumerate(imgs):
        x.tofile(tmp_path / "calib" / "input" / f"{i}.bin")
    args = types.SimpleNamespace(input_dir=str(tmp_path / "calib"), data_num=6, rank=0, local_rank=0, world_size=1,
                                 deploy="trt", calib_batch=4)
    # record, on the host, exactly the activations the calibration forward produced (batch 4 then batch 2): OCTAV is then
    # held to the north-star tolerance, 1e-5, instead of absorbing the difference between a batched and a batch-1 forward
    from dipoorlet_amd.executor import GraphSession
    rec = {}
    orig = GraphSession.run

    def spy(self, inputs):
        res = orig(self, inputs)
        for n, t in zip(self.tensor_names, res):
            rec.setdefault(n, []).extend(x.reshape(-1) for x in t.cpu().numpy())
        return res
    GraphSession.run = spy
    try:
        stats = forward_net_octav(g, args)
    finally:
        GraphSession.run = orig
    assert all(len(v) == 6 for v in rec.values())
    for n, per_img in rec.items():
        for i, x in enumerate(per_img):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref_s = O.octav_scale(x, 1)
            assert np.isclose(stats[n]["optimal_s"][i], ref_s, rtol=1e-5, atol=1e-5), (n, i, stats[n]["optimal_s"][i], ref_s)
            assert stats[n]["max"][i] == x.max() and stats[n]["min"][i] == x.min(), (n, i)
    rc = main(["-M", str(tmp_path / "vit.onnx"), "-I", str(tmp_path / "calib"), "-N", "6", "-A", "mse", "-D", "trt", "-O",
               str(tmp_path / "out"), "--calib_batch", "4", "--bc"])
    assert rc == 0
    assert os.path.exists(tmp_path / "out" / "update_bias_model.onnx") and os.path.exists(tmp_path / "out" / "trt_clip_val.json")


@pytest.mark.two_forwards
def test_activation_cache_by_name(workdir, activations):
    import types

    from dipoorlet_amd.forward_net import ActivationCache
    g, acts = activations
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, calib_batch=BATCH)
    cache = ActivationCache(g, args, 2, 7)
    t = cache["layer1.0.add_out"]
    assert len(t) == 5 and t[0].is_cuda and tuple(t[0].shape) == tuple(g.get_tensor_shape("layer1.0.add_out")[1:])
    assert np.allclose(t[0].cpu().numpy().ravel(), acts["layer1.0.add_out"][2], rtol=1e-4, atol=1e-5)
    assert cache["conv1.weight"].shape == (64, 3, 7, 7)
    cache.reset()
    assert not cache.activation_cache


@pytest.mark.parametrize("deploy", ["trt", "snpe", "ti"])
def test_session_folds_all_weight_fake_quants_in_one_launch(workdir, deploy):
    """A quant_graph session quantises every weight once at build time, all of them in ONE dpl_fake_quant_items launch
    (executor.GraphSession._fold_weights): bit-identical to one QDQNode.apply per weight (per-channel rows on the weight's own
    axis for trt, per-tensor asymmetric grids for snpe, power-of-two scales for ti) and to the oracle's Q -> DQ."""
    import types

    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import tensor_calibration
    from dipoorlet_amd.utils import load_clip_val, save_clip_val
    dist_helper.init_default()
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    out = workdir / f"fold_{deploy}"
    os.makedirs(out, exist_ok=True)
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, rank=0, local_rank=0, world_size=1,
                                 bins=2048, threshold=0.99999, deploy=deploy, act_quant="minmax", calib_batch=BATCH,
                                 output_dir=str(out), skip_layers=[], savefp=False)
    a, w = tensor_calibration(g, args)
    save_clip_val(a, w, args)
    a, w = load_clip_val(args)
    clip = dict(a)
    clip.update(w)
    gq, _ = quant_graph(g, clip, args)
    sq = gq.make_session()
    n = 0
    for node in gq.graph.node:
        if node.name not in sq._folded:
            continue
        q = gq._qdq[node.name]
        x = sq.consts[node.input[0]]
        one = q.apply(x.contiguous())
        assert torch.equal(sq.consts[node.output[0]], one), node.name
        want = O.fake_quant_qdq(x.cpu().numpy(), q.scale, q.zero_point_as_stored(), axis=q.axis if q.scale.size > 1 else None, signed=q.symmetric)
        assert np.array_equal(one.cpu().numpy(), want), node.name
        n += 1
    assert n >= 20


@pytest.mark.two_forwards
def test_session_warms_up_for_the_callers_first_batch(workdir):
    """GraphSession(first_batch=...): the session asks how many images the caller's first forward carries as soon as the shapes
    are known — BEFORE the weights travel — and starts the convolutions' first calls for that batch size on zero weights; what it
    computes is what a session built without the hook computes, bit for bit (the library's deterministic algorithms: by default
    the 3 x 3 stride-2 convolutions add split partial sums with fp32 atomics and no two forwards agree to the last bit,
    pre-warmed or not — conftest).  A quantised graph (convolution weights behind folded FakeQuant nodes) warms up the same way."""
    import types

    from dipoorlet_amd import dist_helper, executor
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.utils import MARKS
    dist_helper.init_default()
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    asked = []

    def first_batch(sess):
        asked.append((len(sess.tensor_names), len(sess.consts)))      # shapes known, nothing on the device yet
        return 4
    plain = g.make_session()
    MARKS.pop("session:conv_threads_started", None)
    MARKS.pop("session:consts_issued", None)
    warm = g.make_session(first_batch=first_batch)
    assert asked == [(len(plain.tensor_names), 0)] and warm._prewarmed
    assert MARKS["session:conv_threads_started"] <= MARKS["session:consts_issued"]
    x = {n: torch.randn([4] + [int(d) for d in g.get_tensor_shape(n)[1:]], device="cuda") for n in plain.input_names}
    assert all(torch.equal(u, v) for u, v in zip(plain.run(x), warm.run(x)))
    executor.join_helpers()
    clip = {n: [-3.0, 3.0] for n in plain.tensor_names}
    from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight
    clip.update(find_clip_val_minmax_weight(g, None, session=plain))
    gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
    q1, q2 = gq.make_session(), gq.make_session(first_batch=lambda s: 4)
    assert q2._prewarmed and q2._folded == q1._folded and len(q2._folded) > 0
    for a, b in zip(q1.run(x), q2.run(x)):
        assert torch.equal(a, b)
    executor.join_helpers()


@pytest.mark.two_forwards
def test_relu_and_add_relu_fused_into_the_fake_quant_kernel(workdir, monkeypatch):
    """executor.relu_fusion: a fake-quantised session asked for the network output only runs ReLU -> Q/DQ and Add -> ReLU -> Q/DQ
    chains as ONE k_fake_quant<PRE> launch each (the merge-ReLU rule, quantize.py:50-55, puts the next layer's Q/DQ pair directly
    behind that ReLU); the output is bit for bit the one of the session that runs every node on its own (run(): every tensor
    exposed), a tensor asked for by name is never fused away, and --bc — whose node-major walk fuses the same chains — writes the
    same biases with DPL_FUSE_RELU=0."""
    import types

    from dipoorlet_amd import dist_helper, executor
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.quantize import quant_graph
    from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight, tensor_calibration
    from dipoorlet_amd.weight_transform import bias_correction
    dist_helper.init_default()
    g = ONNXGraph.load(str(workdir / "model.onnx"))
    plain = g.make_session()
    clip = {n: [-3.0, 3.0] for n in plain.tensor_names}
    clip.update(find_clip_val_minmax_weight(g, None, session=plain))
    gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
    sq = gq.make_session()
    out = gq.network_outputs[0]
    fused, skipped = sq.fusion([out])
    kinds = [p for p, _ in fused.values()]
    assert kinds.count("relu") >= 8 and kinds.count("add_relu") >= 6 and len(skipped) == kinds.count("relu") + 2 * kinds.count("add_relu")
    x = {n: torch.randn([4] + [int(d) for d in g.get_tensor_shape(n)[1:]], device="cuda") for n in plain.input_names}
    launches = []
    orig = executor.fused_fake_quant
    monkeypatch.setattr(executor, "fused_fake_quant", lambda s_, node, pre, *xs: (launches.append(pre), orig(s_, node, pre, *xs))[1])
    every = dict(zip(sq.tensor_names, sq.run(x)))                    # every node on its own
    assert launches == []
    (y,) = sq.run_named(x, [out])                                    # fused
    assert sorted(launches) == sorted(kinds) and torch.equal(y, every[out])
    # a ReLU output asked for by name is computed (and equals the unfused one); the other chains stay fused
    relu_out = next(n.output[0] for n in gq.graph.node if n.name in skipped and n.op_type == "Relu")
    launches.clear()
    y2, r2 = sq.run_named(x, [out, relu_out])
    assert torch.equal(y2, every[out]) and torch.equal(r2, every[relu_out]) and len(launches) == len(kinds) - 1
    monkeypatch.undo()
    # --bc with and without the fusion
    o = workdir / "bc_fuse"
    os.makedirs(o, exist_ok=True)
    args = types.SimpleNamespace(input_dir=str(workdir / "calib"), data_num=N, rank=0, local_rank=0, world_size=1, bins=2048,
                                 threshold=0.99999, deploy="trt", act_quant="minmax", calib_batch=BATCH, output_dir=str(o), skip_layers=[])
    a, w = tensor_calibration(g, args)
    g_fused = bias_correction(g, a, w, args)
    monkeypatch.setenv("DPL_FUSE_RELU", "0")
    g_plain = bias_correction(g, a, w, args)
    for node in g.graph.node:
        if node.op_type in ("Conv", "Gemm"):
            b = next(n for n in g_fused.graph.node if n.name == node.name).input[2]
            assert np.array_equal(g_fused.get_initializer(b), g_plain.get_initializer(b)), node.name
