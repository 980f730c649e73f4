"""Builders of the BASELINE networks as ONNX graphs with seeded random weights (no checkpoints or
torchvision offline): ResNet-18 / ResNet-50 in the form `onnxsim` leaves them (BatchNorm folded into
Conv + bias, torchvision v1.5 topology) and ViT-B/16 in the decomposed form torch exports.

They exist so that the BASELINE configs run from a real `.onnx` through the package's own reader and
executor; tensor counts match SURVEY §8: ResNet-50 T = 123 / 26,598,376 elems, ResNet-18 T = 50 / 5,897,704.
"""
import numpy as np

from . import onnx_io
from .graph import ONNXGraph
from .onnx_io import Node


class _B:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.nodes, self.init = [], {}
        self.k = 0

    def w(self, name, shape, fan_in=None, scale=1.0):
        fan_in = fan_in or int(np.prod(shape[1:])) or 1
        a = (self.rng.standard_normal(shape) * np.sqrt(2.0 / fan_in) * scale).astype(np.float32)
        self.init[name] = a
        return name

    def b(self, name, n, scale=0.05):
        self.init[name] = (self.rng.standard_normal(n) * scale).astype(np.float32)
        return name

    def const(self, name, arr):
        self.init[name] = np.asarray(arr)
        return name

    def node(self, op, inputs, out=None, **attrs):
        self.k += 1
        name = out or f"{op.lower()}_{self.k}"
        self.nodes.append(Node(op, inputs, [name], name=f"{op}_{self.k}", attrs=attrs))
        return name

    def conv(self, x, cin, cout, k, stride=1, pad=0, tag=None, groups=1, scale=1.0):
        tag = tag or f"conv{self.k}"
        w = self.w(f"{tag}.weight", (cout, cin // groups, k, k), scale=scale)
        b = self.b(f"{tag}.bias", cout)
        return self.node("Conv", [x, w, b], out=f"{tag}_out", dilations=[1, 1], group=groups, kernel_shape=[k, k],
                         pads=[pad] * 4, strides=[stride, stride])

    def finish(self, inp, inp_shape, out):
        m = onnx_io.Model()
        m.nodes, m.initializers = self.nodes, self.init
        m.inputs = [(inp, onnx_io.FLOAT, list(inp_shape))]
        m.outputs = [(out, onnx_io.FLOAT, None)]
        m.producer_name = "dipoorlet_amd.models"
        return ONNXGraph(m)


def _resnet(block, layers, seed, num_classes=1000, width=64, image=224):
    g = _B(seed)
    x = g.conv("input", 3, width, 7, 2, 3, "conv1")
    x = g.node("Relu", [x], out="relu1_out")
    x = g.node("MaxPool", [x], out="maxpool_out", ceil_mode=0, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2])
    cin = width
    for li, nb in enumerate(layers):
        w = width * 2 ** li
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            p = f"layer{li + 1}.{bi}"
            idt = x
            if block == "bottleneck":
                cout = 4 * w
                y = g.node("Relu", [g.conv(x, cin, w, 1, 1, 0, p + ".conv1")], out=p + ".relu1_out")
                y = g.node("Relu", [g.conv(y, w, w, 3, stride, 1, p + ".conv2")], out=p + ".relu2_out")
                y = g.conv(y, w, cout, 1, 1, 0, p + ".conv3", scale=0.3)  # keeps the residual stream bounded
            else:
                cout = w
                y = g.node("Relu", [g.conv(x, cin, w, 3, stride, 1, p + ".conv1")], out=p + ".relu1_out")
                y = g.conv(y, w, w, 3, 1, 1, p + ".conv2", scale=0.3)
            if stride != 1 or cin != cout:
                idt = g.conv(x, cin, cout, 1, stride, 0, p + ".downsample")
            y = g.node("Add", [y, idt], out=p + ".add_out")
            x = g.node("Relu", [y], out=p + ".relu_out")
            cin = cout
    x = g.node("GlobalAveragePool", [x], out="avgpool_out")
    x = g.node("Flatten", [x], out="flatten_out", axis=1)
    wfc = g.w("fc.weight", (num_classes, cin))
    bfc = g.b("fc.bias", num_classes)
    x = g.node("Gemm", [x, wfc, bfc], out="output", alpha=1.0, beta=1.0, transB=1)
    return g.finish("input", (1, 3, image, image), x)


def resnet18(seed=0, **kw):
    return _resnet("basic", (2, 2, 2, 2), seed, **kw)


def resnet50(seed=0, **kw):
    return _resnet("bottleneck", (3, 4, 6, 3), seed, **kw)


def vit(seed=0, image=224, patch=16, dim=768, depth=12, heads=12, mlp=3072, num_classes=1000, attn_gain=1.0):
    """ViT-B/16 with decomposed LayerNorm / attention / erf-GELU, as the TorchScript exporter emits them.
    attn_gain scales the attention logits: random weights give near-uniform attention rows, a trained network peaked ones
    (most probabilities far below 2^-18); a gain of 8 .. 12 makes the random network's softmax outputs look like the latter."""
    g = _B(seed)
    n_tok = (image // patch) ** 2
    hd = dim // heads
    x = g.conv("input", 3, dim, patch, patch, 0, "patch_embed")
    x = g.node("Reshape", [x, g.const("shape_tokens", np.array([1, dim, n_tok], np.int64))])
    x = g.node("Transpose", [x], perm=[0, 2, 1])
    cls = g.const("cls_token", (g.rng.standard_normal((1, 1, dim)) * 0.02).astype(np.float32))
    x = g.node("Concat", [cls, x], axis=1)
    pos = g.const("pos_embed", (g.rng.standard_normal((1, n_tok + 1, dim)) * 0.02).astype(np.float32))
    x = g.node("Add", [x, pos])

    def layer_norm(x, tag):
        mu = g.node("ReduceMean", [x], axes=[-1], keepdims=1)
        d = g.node("Sub", [x, mu])
        var = g.node("ReduceMean", [g.node("Mul", [d, d])], axes=[-1], keepdims=1)
        sd = g.node("Sqrt", [g.node("Add", [var, g.const(tag + ".eps", np.float32(1e-6))])])
        y = g.node("Div", [d, sd])
        y = g.node("Mul", [y, g.const(tag + ".weight", (1 + 0.1 * g.rng.standard_normal(dim)).astype(np.float32))])
        return g.node("Add", [y, g.b(tag + ".bias", dim)])

    def linear(x, cin, cout, tag):
        w = g.w(tag + ".weight", (cin, cout), fan_in=cin, scale=0.7)
        return g.node("Add", [g.node("MatMul", [x, w]), g.b(tag + ".bias", cout)])

    for i in range(depth):
        p = f"blocks.{i}"
        y = layer_norm(x, p + ".norm1")
        qkv = linear(y, dim, 3 * dim, p + ".attn.qkv")
        qkv = g.node("Reshape", [qkv, g.const(p + ".shape_qkv", np.array([1, n_tok + 1, 3, heads, hd], np.int64))])
        qkv = g.node("Transpose", [qkv], perm=[2, 0, 3, 1, 4])
        q = g.node("Gather", [qkv, g.const(p + ".i0", np.array(0, np.int64))], axis=0)
        k = g.node("Gather", [qkv, g.const(p + ".i1", np.array(1, np.int64))], axis=0)
        v = g.node("Gather", [qkv, g.const(p + ".i2", np.array(2, np.int64))], axis=0)
        att = g.node("MatMul", [q, g.node("Transpose", [k], perm=[0, 1, 3, 2])])
        att = g.node("Mul", [att, g.const(p + ".scale", np.float32(hd ** -0.5 * attn_gain))])
        att = g.node("Softmax", [att], axis=-1)
        y = g.node("MatMul", [att, v])
        y = g.node("Transpose", [y], perm=[0, 2, 1, 3])
        y = g.node("Reshape", [y, g.const(p + ".shape_out", np.array([1, n_tok + 1, dim], np.int64))])
        y = linear(y, dim, dim, p + ".attn.proj")
        x = g.node("Add", [x, y])
        y = layer_norm(x, p + ".norm2")
        y = linear(y, dim, mlp, p + ".mlp.fc1")
        e = g.node("Erf", [g.node("Div", [y, g.const(p + ".sqrt2", np.float32(np.sqrt(2.0)))])])
        y = g.node("Mul", [g.node("Mul", [y, g.node("Add", [e, g.const(p + ".one", np.float32(1.0))])]),
                           g.const(p + ".half", np.float32(0.5))])
        y = linear(y, mlp, dim, p + ".mlp.fc2")
        x = g.node("Add", [x, y])
    x = layer_norm(x, "norm")
    x = g.node("Gather", [x, g.const("cls_index", np.array(0, np.int64))], axis=1)
    wfc = g.w("head.weight", (num_classes, dim))
    x = g.node("Gemm", [x, wfc, g.b("head.bias", num_classes)], out="output", alpha=1.0, beta=1.0, transB=1)
    return g.finish("input", (1, 3, image, image), x)


def vit_b16(seed=0, **kw):
    return vit(seed, **kw)
