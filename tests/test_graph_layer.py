"""CPU: the graph layer that stands in for onnx / onnxruntime on the calibration path — wire-format
reader/writer, ONNXGraph surface, torch executor (all node outputs, batched)."""
import os
import warnings

import numpy as np
import pytest
import torch

from dipoorlet_amd import models, onnx_io
from dipoorlet_amd.executor import GraphSession
from dipoorlet_amd.graph import ONNXGraph


def _torch_export(model, x, path):
    """Real ONNX bytes from torch's C++ serializer (TorchScript exporter internals; the public entry point
    insists on the absent `onnx` package only for a post-processing step that is a no-op here)."""
    try:
        from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as P
        from torch.onnx._internal.torchscript_exporter import utils as U
    except Exception as e:  # pragma: no cover
        pytest.skip(f"torch internal exporter unavailable: {e}")
    P._add_onnxscript_fn = lambda proto, custom_opsets: proto
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U._export(model, (x,), path, opset_version=13, input_names=["input"], output_names=["out"])


class SmallCNN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(8)
        self.c2 = torch.nn.Conv2d(8, 8, 3, stride=2, padding=1, groups=2)
        self.fc = torch.nn.Linear(8 * 8 * 8, 10)

    def forward(self, x):
        y = torch.relu(self.bn(self.c1(x)))
        z = torch.nn.functional.max_pool2d(y, 3, 2, 1)
        y = torch.nn.functional.avg_pool2d(torch.sigmoid(self.c2(y)), 1) + z
        return self.fc(y.flatten(1))


class SmallMLP(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l1 = torch.nn.Linear(16, 32)
        self.ln = torch.nn.LayerNorm(32)
        self.l2 = torch.nn.Linear(32, 8)

    def forward(self, x):
        y = torch.nn.functional.gelu(self.ln(self.l1(x)))
        return torch.softmax(self.l2(y), -1)


@pytest.mark.parametrize("mk,shape", [("SmallCNN", (1, 3, 16, 16)), ("SmallMLP", (1, 5, 16)), ("ShapeZoo", (1, 3, 17, 19))])
def test_reader_and_executor_on_torch_exported_onnx(tmp_path, mk, shape):
    mk = globals()[mk]
    torch.manual_seed(0)
    model = mk().eval()
    for p in model.parameters():
        p.data.normal_(0, 0.3)
    x = torch.randn(*shape)
    path = str(tmp_path / "m.onnx")
    _torch_export(model, x, path)
    g = ONNXGraph.load(path)
    assert g.network_inputs == ["input"] and g.network_outputs == ["out"]
    assert g.get_tensor_shape("input") == list(shape)
    s = GraphSession(g, device="cpu")
    assert s.tensor_names[0] == "input" and "out" in s.tensor_names
    # single image
    got = dict(zip(s.tensor_names, s.run({"input": x})))
    ref = model(x)
    assert torch.allclose(got["out"], ref, rtol=1e-4, atol=1e-5)
    # batch of 3 stacked on dim 0 == three single runs
    xb = torch.randn(3 * shape[0], *shape[1:])
    gb = dict(zip(s.tensor_names, s.run({"input": xb})))
    assert torch.allclose(gb["out"].reshape(3, -1), model(xb).reshape(3, -1), rtol=1e-4, atol=1e-5)
    for n, e in zip(s.tensor_names, s.elems_per_image):
        assert gb[n].shape[0] == 3 and gb[n].numel() == 3 * e and gb[n].is_contiguous()
    # every node output is exposed, network inputs first (forward_net.py:193-198, 220-235)
    outs = [o for n in g.graph.node for o in n.output]
    if mk is ShapeZoo:      # (its integer tensors — the exported shape arithmetic — are not calibration tensors)
        assert s.tensor_names == [n for n in ["input"] + outs if n in set(s.tensor_names)] and len(s.tensor_names) > 20
    else:
        assert s.tensor_names == ["input"] + outs


def test_writer_roundtrip_and_reference_graph_surface(tmp_path):
    g = models.resnet18(seed=3)
    g.output_dir = str(tmp_path)
    path = g.save_onnx_model("r18")
    assert os.path.getsize(path) > 40e6
    g2 = ONNXGraph.load(path)
    assert [(n.op_type, n.name, n.input, n.output, n.attrs) for n in g2.graph.node] == \
           [(n.op_type, n.name, n.input, n.output, n.attrs) for n in g.graph.node]
    assert set(g2.initializer) == set(g.initializer)
    assert all(np.array_equal(g2.initializer[k], v) for k, v in g.initializer.items())
    assert g2.network_inputs == ["input"] and g2.network_outputs == ["output"]
    conv1 = g2.graph.node[0]
    assert g2.get_tensor_producer("input") == "INPUT_TOKEN" and g2.get_tensor_producer("conv1_out") is conv1
    assert g2.get_tensor_consumer("output") == ["OUTPUT_TOKEN"]
    assert [n.op_type for n in g2.get_tensor_consumer("maxpool_out")] == ["Conv", "Add"]
    assert g2.get_initializer("conv1.weight").shape == (64, 3, 7, 7) and g2.index(conv1) == 0
    assert g2.get_tensor_shape("conv1.weight") == [64, 3, 7, 7]


def test_baseline_network_tensor_sets_match_survey():
    s18 = GraphSession(models.resnet18(), device="cpu")
    s50 = GraphSession(models.resnet50(), device="cpu")
    assert (len(s18.tensor_names), sum(s18.elems_per_image)) == (50, 5897704)
    assert (len(s50.tensor_names), sum(s50.elems_per_image)) == (123, 26598376)
    from dipoorlet_amd.synthetic import resnet50_tensors
    assert sorted(s50.elems_per_image) == sorted(e for _, e, _ in resnet50_tensors())


def test_vit_batch_axis_normalisation():
    s = GraphSession(models.vit(depth=1, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10), device="cpu")
    torch.manual_seed(1)
    x = torch.randn(3, 3, 32, 32)
    a = s.run({"input": x})
    b = s.run({"input": x[2:3]})
    for n, u, v in zip(s.tensor_names, a, b):
        assert u.shape[0] == 3 and tuple(u.shape[1:]) == tuple(v.shape[1:]) or u[2].numel() == v.numel(), n
        assert torch.allclose(u[2].reshape(-1), v.reshape(-1), rtol=1e-4, atol=1e-5), n


def test_wire_format_scalar_and_attr_types(tmp_path):
    m = onnx_io.Model()
    m.nodes = [onnx_io.Node("Foo", ["a", ""], ["b"], name="n0",
                            attrs={"f": 1.5, "i": -3, "s": "txt", "ints": [1, -2, 3], "floats": [0.5, 2.0],
                                   "t": np.arange(6, dtype=np.int64).reshape(2, 3), "strs": ["x", "y"]})]
    m.initializers = {"w": np.float32(3.25).reshape(()), "h": np.arange(4, dtype=np.float16), "u": np.array([1, 255], np.uint8)}
    m.inputs = [("a", onnx_io.FLOAT, [1, 0, 3])]
    m.outputs = [("b", onnx_io.FLOAT, None)]
    p = str(tmp_path / "x.onnx")
    onnx_io.save_model(m, p)
    r = onnx_io.load_model(p)
    n = r.nodes[0]
    assert (n.op_type, n.input, n.output, n.name) == ("Foo", ["a", ""], ["b"], "n0")
    assert n.attrs["f"] == 1.5 and n.attrs["i"] == -3 and n.attrs["s"] == "txt" and n.attrs["ints"] == [1, -2, 3]
    assert n.attrs["floats"] == [0.5, 2.0] and n.attrs["strs"] == ["x", "y"] and np.array_equal(n.attrs["t"], m.nodes[0].attrs["t"])
    assert r.initializers["w"].shape == () and r.initializers["w"] == np.float32(3.25)
    assert r.initializers["h"].dtype == np.float16 and r.initializers["u"].tolist() == [1, 255]
    assert r.inputs == [("a", 1, [1, 0, 3])]


def test_fold_batchnorm_keeps_the_function():
    """ONNXGraph.fold_batchnorm (the Conv + BN fusion onnxsim performs for the reference): same outputs, no BN left,
    a bias is created where the layer had none, a BN with a second consumer of its input is left alone."""
    from dipoorlet_amd.models import _B
    g = _B(9)

    def bn(x, c, tag):
        for nm, arr in ((f"{tag}.g", np.abs(g.rng.standard_normal(c)) + 0.5), (f"{tag}.b", g.rng.standard_normal(c) * 0.1),
                        (f"{tag}.m", g.rng.standard_normal(c) * 0.2), (f"{tag}.v", np.abs(g.rng.standard_normal(c)) + 0.3)):
            g.const(nm, arr.astype(np.float32))
        return g.node("BatchNormalization", [x, f"{tag}.g", f"{tag}.b", f"{tag}.m", f"{tag}.v"], out=f"{tag}_out",
                      epsilon=1e-3)
    x = bn(g.conv("input", 3, 8, 3, 1, 1, "c1"), 8, "bn1")
    x = g.node("Relu", [x], out="r1")
    w = g.w("c2.weight", (6, 8, 1, 1))                                   # a Conv without bias
    x = g.node("Conv", [x, w], out="c2_out", dilations=[1, 1], group=1, kernel_shape=[1, 1], pads=[0] * 4, strides=[1, 1])
    x = bn(x, 6, "bn2")
    side = g.conv(x, 6, 6, 1, 1, 0, "c3")                                # c3_out feeds a BN and an Add: not foldable
    y = bn(side, 6, "bn3")
    x = g.node("Add", [y, side], out="sum")
    x = g.node("GlobalAveragePool", [x], out="gap")
    x = g.node("Flatten", [x], out="flat", axis=1)
    fw = g.w("fc.weight", (5, 6))
    fb = g.b("fc.bias", 5)
    x = bn_in = g.node("Gemm", [x, fw, fb], out="fc_out", transB=1)
    for nm, arr in (("bn4.g", np.ones(5)), ("bn4.b", np.zeros(5)), ("bn4.m", g.rng.standard_normal(5) * 0.1), ("bn4.v", np.ones(5) * 0.7)):
        g.const(nm, arr.astype(np.float32))
    x = g.node("BatchNormalization", [bn_in, "bn4.g", "bn4.b", "bn4.m", "bn4.v"], out="output")
    graph = g.finish("input", [1, 3, 12, 12], "output")
    inp = {"input": torch.from_numpy(np.random.default_rng(0).standard_normal((3, 3, 12, 12)).astype(np.float32))}
    ref = GraphSession(graph, device="cpu").run_named(inp, ["output", "sum"])
    assert graph.fold_batchnorm() == 3
    ops_left = [n.op_type for n in graph.graph.node]
    assert ops_left.count("BatchNormalization") == 1 and "bn1.g" not in graph.initializer
    c2 = next(n for n in graph.graph.node if len(n.input) > 1 and n.input[1] == "c2.weight")
    assert len(c2.input) == 3 and c2.input[2] in graph.initializer and c2.output[0] == "bn2_out"
    out = GraphSession(graph, device="cpu").run_named(inp, ["output", "sum"])
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-5, atol=2e-6)
    path = graph.save_onnx_model("/tmp/_folded_bn_test")                # round-trips through the writer / reader
    again = GraphSession(ONNXGraph.load(path), device="cpu").run_named(inp, ["output"])[0]
    np.testing.assert_allclose(again.numpy(), ref[0].numpy(), rtol=2e-5, atol=2e-6)


def test_batched_execution_is_verified_against_per_image():
    """A graph exported for one image whose Reshape carries a literal NON-leading shape ([4, -1]) mixes the images of a
    batch without raising.  The session notices (batch-2 forward != two batch-1 forwards), runs one image at a time from
    then on, and a batched run() / run_named() returns what the per-image runs return; a well-behaved graph keeps batching."""
    from dipoorlet_amd.onnx_io import Node
    g = ONNXGraph()
    g.graph.node = [Node("Relu", ["x"], ["r"], name="relu"),
                    Node("Reshape", ["r", "shp"], ["y"], name="reshape"),
                    Node("Mul", ["y", "k"], ["z"], name="mul")]
    g.initializer = {"shp": np.array([4, -1], np.int64), "k": np.array(2.0, np.float32)}
    g.network_inputs, g.network_outputs = ["x"], ["z"]
    g.input = ["x", "shp", "k"]
    g.tensor_name_shape_map = {"x": [1, 4, 6], "z": [4, 6]}
    g.topologize_graph()
    g.set_index()
    s = GraphSession(g, device="cpu")
    x = torch.randn(3, 4, 6)
    outs = s.run({"x": x})
    assert s._batched_ok is False
    for k in range(3):
        one = s.run({"x": x[k:k + 1]})
        for name, tb, t1 in zip(s.tensor_names, outs, one):
            assert torch.equal(tb[k:k + 1].reshape(t1.shape), t1), name
    named = s.run_named({"x": x}, ["z"])[0]
    assert torch.equal(named, torch.stack([s.run_named({"x": x[k:k + 1]}, ["z"])[0] for k in range(3)]))
    ok = GraphSession(models.resnet18(), device="cpu")
    ok.run({n: torch.randn(2, *ok.graph.get_tensor_shape(n)[1:]) for n in ok.input_names})
    assert ok._batched_ok is True


def _shape_sets(g, monkeypatch):
    """(names, elems, per-image shapes) from the host rules and from the real batch-1 forward."""
    from dipoorlet_amd.forward_net import WALL
    monkeypatch.setenv("DPL_INFER_DEVICE", "0")
    WALL.pop("session_infer_host_s", None)
    a = GraphSession(g, device="cpu")
    assert "session_infer_host_s" in WALL and "session_infer_device_s" not in WALL   # no silent fall-back to the forward
    monkeypatch.setenv("DPL_INFER_DEVICE", "1")
    b = GraphSession(g, device="cpu")
    WALL.pop("session_infer_device_s", None)
    return (a.tensor_names, a.elems_per_image, a.shape1), (b.tensor_names, b.elems_per_image, b.shape1)


class ShapeZoo(torch.nn.Module):
    """Ops whose output shape takes arithmetic: strided / dilated / grouped / transposed convolutions, pools in ceil mode,
    pads, slices, concat, upsampling, squeeze / unsqueeze, reductions, a view computed from x.shape (Shape -> Gather ->
    Concat -> Reshape in the exported graph), matmul broadcasting."""

    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 8, 5, stride=2, padding=1, dilation=2)
        self.ct = torch.nn.ConvTranspose2d(8, 6, 3, stride=2, padding=1, output_padding=1)
        self.c3 = torch.nn.Conv2d(6, 6, 3, padding=1, groups=3)
        self.p = torch.nn.PReLU(6)
        self.fc = torch.nn.Linear(14, 7)

    def forward(self, x):
        y = self.c1(x)                                                    # [1, 8, 6, 7] from 17 x 19
        y = torch.nn.functional.max_pool2d(y, 3, 2, 1, ceil_mode=True)
        y = self.p(self.c3(self.ct(y)))
        y = torch.nn.functional.pad(y, (1, 2, 0, 1))
        y = torch.nn.functional.avg_pool2d(y, 2, 2, ceil_mode=True, count_include_pad=True)
        a, b = y[:, :4], y[:, 2:, 1:]
        z = torch.cat([a[:, :, 1:, 1:], b[:, :, :, 1:]], 1)
        z = torch.nn.functional.interpolate(z, scale_factor=2.0, mode="nearest")
        z = z.reshape(z.shape[0], z.shape[1], -1)                          # Shape -> Gather -> ... -> Reshape
        w = z.mean(-1, keepdim=True).squeeze(-1).unsqueeze(1)              # [1, 1, 8]
        m = torch.matmul(z.transpose(1, 2)[:, :14], z[:, :, :14])          # [1, 14, 14]
        return self.fc(m).sum(1) + w.amax(-1)


def test_host_shape_inference_matches_the_forward(tmp_path, monkeypatch):
    """executor.GraphSession takes every tensor's per-image shape from shape_infer's host rules, not from a batch-1
    forward (forward_net.py:193-202 needs none: ORT infers).  Every rule is held to the shapes the real forward produces."""
    for g in (models.resnet18(), models.resnet50(), models.vit(depth=2, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10),
              models.resnet18(image=97)):
        host, real = _shape_sets(g, monkeypatch)
        assert host == real
    torch.manual_seed(0)
    for mk, shape in ((SmallCNN, (1, 3, 16, 16)), (SmallMLP, (1, 5, 16)), (ShapeZoo, (1, 3, 17, 19))):
        path = str(tmp_path / f"{mk.__name__}.onnx")
        _torch_export(mk().eval(), torch.randn(*shape), path)
        g = ONNXGraph.load(path)
        host, real = _shape_sets(g, monkeypatch)
        assert host == real, mk.__name__
        assert len(host[0]) > 5
    # hand-built: Split, Slice with steps, Gather by a vector, Where / Equal / Expand / ConstantOfShape / Cast, asymmetric pool pads
    from dipoorlet_amd.onnx_io import Node
    g = ONNXGraph()
    i64 = lambda *v: np.array(v, np.int64)    # noqa: E731
    g.graph.node = [
        Node("Split", ["x", "sizes"], ["s0", "s1"], name="split", attrs={"axis": 1}),
        Node("Slice", ["s1", "st", "en", "ax", "sp"], ["sl"], name="slice"),
        Node("Gather", ["sl", "gi"], ["ga"], name="gather", attrs={"axis": 2}),
        Node("MaxPool", ["s0"], ["mp"], name="mp", attrs={"kernel_shape": [2, 2], "strides": [2, 2], "pads": [0, 1, 1, 0]}),
        Node("Shape", ["mp"], ["mps"], name="shape"),
        Node("ConstantOfShape", ["mps"], ["zeros"], name="cos", attrs={"value": np.zeros(1, np.float32)}),
        Node("Equal", ["mp", "zeros"], ["eq"], name="eq"),
        Node("Where", ["eq", "one", "mp"], ["wh"], name="where"),
        Node("Cast", ["eq"], ["eqf"], name="cast", attrs={"to": 1}),
        Node("Expand", ["k", "mps"], ["ex"], name="expand"),
        Node("Add", ["wh", "ex"], ["y"], name="add"),
        Node("ReduceMax", ["y"], ["rm"], name="rmax", attrs={"axes": [2, 3], "keepdims": 0}),
        Node("GlobalMaxPool", ["ga"], ["gm"], name="gmp"),
    ]
    g.initializer = {"sizes": i64(2, 3), "st": i64(1, 0), "en": i64(100, -1), "ax": i64(2, 3), "sp": i64(2, 1), "gi": i64(0, 2, 1, 0),
                     "one": np.array(1.0, np.float32), "k": np.full((1, 1, 1), 0.5, np.float32)}
    g.network_inputs, g.network_outputs = ["x"], ["rm", "gm"]
    g.input = ["x"] + list(g.initializer)
    g.tensor_name_shape_map = {"x": [1, 5, 9, 8]}
    g.topologize_graph()
    g.set_index()
    host, real = _shape_sets(g, monkeypatch)
    assert host == real
    assert dict(zip(host[0], host[1]))["ga"] == 3 * 4 * 7       # (the int / bool tensors are not calibration tensors)
    assert "eq" not in host[0] and "mps" not in host[0] and "eqf" in host[0]


def test_host_shape_inference_falls_back_when_a_rule_is_missing(monkeypatch):
    from dipoorlet_amd import shape_infer
    from dipoorlet_amd.forward_net import WALL
    monkeypatch.delitem(shape_infer._RULES, "MaxPool")
    monkeypatch.setenv("DPL_INFER_DEVICE", "0")
    WALL.pop("session_infer_device_s", None)
    s = GraphSession(models.resnet18(), device="cpu")
    assert "session_infer_device_s" in WALL and (len(s.tensor_names), sum(s.elems_per_image)) == (50, 5897704)


def test_static_batching_proof_never_contradicts_the_dynamic_check(tmp_path, monkeypatch):
    """shape_infer.batch_transparent lets a session skip the batch-2-against-batch-1 verification.  Whenever it says 'proven',
    the verification — forced here — must agree; the graph that scrambles a batch must not be proven."""
    graphs = [("r18", models.resnet18()), ("vit", models.vit(depth=2, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10))]
    torch.manual_seed(0)
    for mk, shape in ((SmallCNN, (1, 3, 16, 16)), (SmallMLP, (1, 5, 16)), (ShapeZoo, (1, 3, 17, 19))):
        path = str(tmp_path / f"{mk.__name__}.onnx")
        _torch_export(mk().eval(), torch.randn(*shape), path)
        graphs.append((mk.__name__, ONNXGraph.load(path)))
    proven = {}
    for name, g in graphs:
        monkeypatch.setenv("DPL_EXECUTOR_VERIFY_BATCHING", "0")
        proven[name] = GraphSession(g, device="cpu")._batched_ok
        monkeypatch.setenv("DPL_EXECUTOR_VERIFY_BATCHING", "1")
        s = GraphSession(g, device="cpu")
        assert s._batched_ok is None
        assert s.batched_ok() or not proven[name], name
    assert proven["r18"] is True and proven["vit"] is True and proven["SmallCNN"] is True
    # the scrambling Reshape of test_batched_execution_is_verified_against_per_image: no proof
    from dipoorlet_amd.onnx_io import Node
    monkeypatch.setenv("DPL_EXECUTOR_VERIFY_BATCHING", "0")
    g = ONNXGraph()
    g.graph.node = [Node("Relu", ["x"], ["r"], name="relu"), Node("Reshape", ["r", "shp"], ["y"], name="reshape")]
    g.initializer = {"shp": np.array([4, -1], np.int64)}
    g.network_inputs, g.network_outputs = ["x"], ["y"]
    g.input = ["x", "shp"]
    g.tensor_name_shape_map = {"x": [1, 4, 6]}
    g.topologize_graph()
    g.set_index()
    assert GraphSession(g, device="cpu")._batched_ok is None
    # ... nor an axis-0 Concat of the input with itself, a Transpose that is followed by an NCHW op, a reduction over the batch
    for nodes, init in (([Node("Concat", ["x", "x"], ["y"], name="c", attrs={"axis": 0})], {}),
                        ([Node("Transpose", ["x"], ["t"], name="t", attrs={"perm": [1, 0, 2]}),
                          Node("GlobalAveragePool", ["t"], ["y"], name="gap")], {}),
                        ([Node("ReduceMean", ["x"], ["y"], name="rm", attrs={"axes": [0], "keepdims": 1})], {}),
                        ([Node("Add", ["x", "k"], ["y"], name="add")], {"k": np.ones((2, 4, 6), np.float32)})):
        g = ONNXGraph()
        g.graph.node, g.initializer = nodes, dict(init)
        g.network_inputs, g.network_outputs = ["x"], ["y"]
        g.input = ["x"] + list(init)
        g.tensor_name_shape_map = {"x": [1, 4, 6]}
        g.topologize_graph()
        g.set_index()
        from dipoorlet_amd import shape_infer
        env = shape_infer.infer(g, set(), ["x"], 1)
        assert shape_infer.batch_transparent(g, set(), ["x"], env) is False, nodes[0].op_type


def test_session_knows_whether_a_forward_needs_the_blas_library():
    """GraphSession.needs_blas (decides whether a fresh process loads hipBLASLt at all): a convolutional network's classifier head
    is a small product at calibration batch sizes, a transformer's attention is not (activation x activation)."""
    from dipoorlet_amd import models
    from dipoorlet_amd.executor import GraphSession, small_gemm
    s = GraphSession(models.resnet50(), device="cpu")
    assert not s.needs_blas(64) and not s.needs_blas(1) and s.needs_blas(4096)
    assert small_gemm(64, 1000, 2048) and not small_gemm(64 * 197, 3072, 768)
    v = GraphSession(models.vit(depth=2, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10), device="cpu")
    assert v.needs_blas(1)



def test_scan_op_types_reads_no_tensors(tmp_path):
    """onnx_io.scan_op_types: the op types of a model file counted from the mapped file, tensors stepped over — the same counts as a
    full load gives."""
    from collections import Counter
    from dipoorlet_amd import models, onnx_io
    for g, name in ((models.resnet18(), "r18"), (models.vit(depth=2, dim=64, heads=4, mlp=128, image=32, patch=8, num_classes=10), "vit")):
        g.output_dir = str(tmp_path)
        g.save_onnx_model(name)
        path = os.path.join(str(tmp_path), name + ".onnx")
        assert onnx_io.scan_op_types(path) == dict(Counter(n.op_type for n in onnx_io.load_model(path).nodes))


def test_negative_step_slice_follows_onnx_for_empty_ranges():
    """Slice with a negative step (the exporter reverses F.pad's vector this way): clamped bounds as ONNX defines them, and an
    EMPTY result where start <= end — what shape_infer's rule predicts — instead of torch.arange's refusal."""
    import types

    from dipoorlet_amd import executor, shape_infer
    node = types.SimpleNamespace(attrs={}, op_type="Slice")
    x = torch.arange(10.0).reshape(2, 5)
    t = torch.tensor
    run = executor._OPS["Slice"]
    assert run(None, node, x, t([3]), t([0]), t([1]), t([-1])).tolist() == [[3.0, 2.0, 1.0], [8.0, 7.0, 6.0]]
    assert run(None, node, x, t([-1]), t([-6]), t([1]), t([-1])).tolist() == [[4.0, 3.0, 2.0, 1.0, 0.0], [9.0, 8.0, 7.0, 6.0, 5.0]]
    assert run(None, node, x, t([4]), t([-100]), t([1]), t([-2])).tolist() == [[4.0, 2.0, 0.0], [9.0, 7.0, 5.0]]
    assert tuple(run(None, node, x, t([1]), t([3]), t([1]), t([-1])).shape) == (2, 0)
    assert tuple(run(None, node, x, t([2]), t([2]), t([1]), t([-1])).shape) == (2, 0)


def test_relu_fusion_plan_follows_the_merge_relu_rule(monkeypatch):
    """executor.relu_fusion on a fake-quantised ResNet-18 (-D trt; host only: the plan is graph logic).  The reference leaves a ReLU
    behind Conv / Gemm / Add unquantised at its input (quantize.py:50-55), so the next layer's Q/DQ pair sits behind that ReLU
    (:74-93): such a ReLU — and the residual Add in front of it — is fused into the pair's kernel when NOTHING else reads it: not
    when it is a network output, asked for by name, or has a second consumer; a session that exposes every tensor fuses nothing;
    DPL_FUSE_RELU=0 switches the plan off."""
    import types

    from dipoorlet_amd import executor, models
    from dipoorlet_amd.quantize import quant_graph
    g = models.resnet18(seed=1, image=32)
    s = executor.GraphSession(g, device="cpu")
    clip = {n: [-3.0, 3.0] for n in s.tensor_names}
    for node in g.graph.node:
        for i in node.input[1:]:
            if i in g.initializer:
                a = np.asarray(g.initializer[i])
                a2 = a.reshape(a.shape[0], -1)
                clip[i] = [a2.min(-1), a2.max(-1)]
    gq, _ = quant_graph(g, clip, types.SimpleNamespace(deploy="trt", skip_layers=[]))
    sq = executor.GraphSession(gq, device="cpu")
    out = gq.network_outputs[0]
    nodes = {n.name: n for n in gq.graph.node}
    producer = {o: n for n in gq.graph.node for o in n.output}
    fused, skipped = executor.relu_fusion(gq, sq._folded, sq.consts, [out], sq.shape1)
    kinds = [p for p, _ in fused.values()]
    assert kinds.count("relu") == 9 and kinds.count("add_relu") == 7 and len(skipped) == 9 + 2 * 7
    for name, (pre, ins) in fused.items():
        q = nodes[name]
        relu = producer[q.input[0]]
        assert q.op_type == "FakeQuant" and relu.op_type == "Relu" and relu.name in skipped
        if pre == "relu":
            assert ins == [relu.input[0]]
        else:
            add = producer[relu.input[0]]
            assert add.op_type == "Add" and add.name in skipped and ins == list(add.input)
            assert sq.shape1[add.output[0]] is not None
    # (the first ReLU too: -D trt quantises the max pool's input, so the pair behind relu1 is its only reader)
    assert next(n for n in gq.graph.node if n.op_type == "Relu").name in skipped
    # asked for by name: that chain is not fused (neither its ReLU nor its Add), the others are
    one = next(nodes[k] for k, (p, _) in fused.items() if p == "add_relu")
    kept = one.input[0]
    f2, s2 = executor.relu_fusion(gq, sq._folded, sq.consts, [out, kept], sq.shape1)
    assert one.name not in f2 and len(f2) == len(fused) - 1 and len(s2) == len(skipped) - 2
    # every tensor exposed (run()): nothing can be fused; and the switch
    f3, s3 = executor.relu_fusion(gq, sq._folded, sq.consts, sq.tensor_names, sq.shape1)
    assert not f3 and not s3
    assert sq.fusion([out])[0].keys() == fused.keys()
    monkeypatch.setenv("DPL_FUSE_RELU", "0")
    assert executor.relu_fusion(gq, sq._folded, sq.consts, [out], sq.shape1) == ({}, set())
