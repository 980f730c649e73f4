"""Per-image tensor shapes of a graph WITHOUT running it — what `onnx.shape_inference` / the batch-1 ORT run give the
reference (dipoorlet/utils.py:88-117, forward_net.py:193-202), on the host.

The executor used to find every tensor's shape by a batch-1 forward on the device: 0.3 s of a fresh process (the
libraries load and choose kernels for a batch size the run never uses).  Here floating-point tensors are `Spec`s (shape +
dtype, no storage) pushed through one shape rule per op; small constants and everything computed from them — the shape
arithmetic of exported graphs (Shape -> Gather -> Concat -> Reshape) — are REAL host tensors run through the executor's own
ops, so a Reshape target is the value the device run would see.  An op without a rule raises `Unsupported`; the caller then
falls back to the device forward.  tests/test_graph_layer.py holds every rule to the real forward's shapes.
"""
import numpy as np
import torch

SMALL = 4096      # constants up to this many elements keep their values (shape arithmetic, scalars)


class Unsupported(Exception):
    pass


class Spec:
    """A tensor's shape and dtype."""
    __slots__ = ("shape", "dtype")

    def __init__(self, shape, dtype=torch.float32):
        self.shape = tuple(int(d) for d in shape)
        self.dtype = dtype

    def dim(self):
        return len(self.shape)

    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    def is_floating_point(self):
        return self.dtype.is_floating_point

    def __repr__(self):
        return f"Spec({list(self.shape)}, {self.dtype})"


def _real(x):
    return isinstance(x, torch.Tensor)


def _shape(x):
    return tuple(x.shape)


def _dtype2(a, b):
    fa = a.dtype.is_floating_point
    fb = b.dtype.is_floating_point
    if fa or fb:
        return a.dtype if fa else b.dtype
    return torch.promote_types(a.dtype, b.dtype)


def _bcast(*shapes):
    """numpy's rule (torch.broadcast_shapes imports torch._refs on its first call: 0.27 s)."""
    return tuple(int(d) for d in np.broadcast_shapes(*[tuple(int(d) for d in sh) for sh in shapes]))


_RULES = {}


def rule(*names):
    def deco(fn):
        for n in names:
            _RULES[n] = fn
        return fn
    return deco


def _values(ex, v, what):
    if v is None:
        return None
    if not _real(v):
        raise Unsupported(f"{what}: needs the VALUES of a tensor only known by shape")
    return ex._ints(v)


@rule("Relu", "LeakyRelu", "Sigmoid", "Tanh", "HardSigmoid", "HardSwish", "Clip", "Gelu", "Erf", "Sqrt", "Exp", "Log", "Abs",
      "Neg", "Reciprocal", "Floor", "Ceil", "Softmax", "BatchNormalization", "LayerNormalization", "InstanceNormalization",
      "Identity", "Dropout", "FakeQuant")
def _same(ex, s, node, x, *rest):
    return Spec(x.shape, x.dtype)


@rule("Add", "Sub", "Mul", "Div", "Pow", "Eltwise", "PRelu")
def _binary(ex, s, node, a, b):
    dt = _dtype2(a, b)
    if node.op_type == "Div" and not dt.is_floating_point:
        dt = torch.float32
    return Spec(_bcast(_shape(a), _shape(b)), dt)


@rule("Where")
def _where(ex, s, node, c, a, b):
    return Spec(_bcast(_shape(c), _shape(a), _shape(b)), _dtype2(a, b))


@rule("Equal")
def _equal(ex, s, node, a, b):
    return Spec(_bcast(_shape(a), _shape(b)), torch.bool)


@rule("Cast")
def _cast(ex, s, node, x):
    return Spec(x.shape, ex._CAST[int(node.attrs["to"])])


def _conv_len(size, k, stride, dil, pb, pe):
    return (size + pb + pe - dil * (k - 1) - 1) // stride + 1


@rule("Conv")
def _conv(ex, s, node, x, w, b=None):
    nd = len(w.shape) - 2
    strides = node.attrs.get("strides", [1] * nd)
    dil = node.attrs.get("dilations", [1] * nd)
    pads = ex._auto_pad(node, x, list(w.shape[2:]), strides, dil)
    sp = [_conv_len(x.shape[2 + i], w.shape[2 + i], strides[i], dil[i], pads[i], pads[nd + i]) for i in range(nd)]
    return Spec((x.shape[0], w.shape[0], *sp), x.dtype)


@rule("ConvTranspose")
def _convt(ex, s, node, x, w, b=None):
    nd = len(w.shape) - 2
    strides = node.attrs.get("strides", [1] * nd)
    dil = node.attrs.get("dilations", [1] * nd)
    pads = node.attrs.get("pads") or [0] * (2 * nd)
    if list(pads[:nd]) != list(pads[nd:]):
        raise NotImplementedError("ConvTranspose with asymmetric pads")
    opad = node.attrs.get("output_padding", [0] * nd)
    sp = [(x.shape[2 + i] - 1) * strides[i] - 2 * pads[i] + dil[i] * (w.shape[2 + i] - 1) + opad[i] + 1 for i in range(nd)]
    return Spec((x.shape[0], w.shape[1] * int(node.attrs.get("group", 1)), *sp), x.dtype)


def _pool(ex, node, x, avg):
    k, nd, strides, dil, pads = ex._pool_args(node, x)
    if avg:
        dil = [1] * nd
    ceil = bool(node.attrs.get("ceil_mode", 0))
    sym = list(pads[:nd]) == list(pads[nd:])
    if avg and not sym and not node.attrs.get("count_include_pad", 0):
        raise NotImplementedError("AveragePool: asymmetric pads with count_include_pad=0")
    out = []
    for i in range(nd):
        size, pb, pe = x.shape[2 + i], pads[i], pads[nd + i]
        if not sym:               # the executor pads first and pools without padding
            size, pb, pe = size + pb + pe, 0, 0
        num = size + pb + pe - dil[i] * (k[i] - 1) - 1
        o = (-(-num // strides[i]) if ceil else num // strides[i]) + 1
        if ceil and (o - 1) * strides[i] >= size + pb:      # the last window must start inside the input or the left padding
            o -= 1
        out.append(o)
    return Spec((x.shape[0], x.shape[1], *out), x.dtype)


@rule("MaxPool")
def _maxpool(ex, s, node, x):
    return _pool(ex, node, x, False)


@rule("AveragePool")
def _avgpool(ex, s, node, x):
    return _pool(ex, node, x, True)


@rule("GlobalAveragePool", "GlobalMaxPool")
def _gpool(ex, s, node, x):
    return Spec(tuple(x.shape[:2]) + (1,) * (len(x.shape) - 2), x.dtype)


def _matmul_shape(a, b):
    a, b = list(a), list(b)
    if not a or not b:
        raise Unsupported("MatMul of a 0-d tensor")
    va, vb = len(a) == 1, len(b) == 1
    if va:
        a = [1] + a
    if vb:
        b = b + [1]
    if a[-1] != b[-2]:
        raise ValueError(f"MatMul: inner dimensions differ ({a} x {b})")
    lead = list(_bcast(tuple(a[:-2]), tuple(b[:-2])))
    out = lead + ([] if va else [a[-2]]) + ([] if vb else [b[-1]])
    return tuple(out)


@rule("MatMul")
def _matmul(ex, s, node, a, b):
    return Spec(_matmul_shape(a.shape, b.shape), _dtype2(a, b))


@rule("Gemm")
def _gemm(ex, s, node, a, b, c=None):
    sa, sb = list(a.shape), list(b.shape)
    if node.attrs.get("transA", 0):
        sa = sa[::-1]
    if node.attrs.get("transB", 0):
        sb = sb[::-1]
    return Spec(_matmul_shape(sa, sb), a.dtype)


@rule("Flatten")
def _flatten(ex, s, node, x):
    ax = int(node.attrs.get("axis", 1))
    ax = ax + len(x.shape) if ax < 0 else ax
    lead = int(np.prod(x.shape[:ax])) if ax > 0 else 1
    n = Spec(x.shape).numel()
    return Spec((lead, n // lead if lead else 0), x.dtype)


@rule("Reshape")
def _reshape(ex, s, node, x, shape):
    shp = _values(ex, shape, "Reshape")
    n = Spec(x.shape).numel()
    if not node.attrs.get("allowzero", 0):
        shp = [x.shape[i] if d == 0 else d for i, d in enumerate(shp)]
    if s.batch > 1 and shp and shp[0] == 1 and x.shape[0] == s.batch:      # (the executor's re-scaling of a literal leading 1)
        want = int(np.prod([d for d in shp if d > 0]))
        if (-1 in shp and n % (want * s.batch) == 0) or (-1 not in shp and want * s.batch == n):
            shp[0] = s.batch
    if shp.count(-1) > 1:
        raise ValueError("Reshape: more than one -1")
    if -1 in shp:
        known = int(np.prod([d for d in shp if d != -1])) if len(shp) > 1 else 1
        if known == 0 or n % known:
            raise ValueError(f"Reshape: {list(x.shape)} does not fit {shp}")
        shp[shp.index(-1)] = n // known
    elif int(np.prod(shp)) != n:
        raise ValueError(f"Reshape: {list(x.shape)} does not fit {shp}")
    return Spec(shp, x.dtype)


@rule("Transpose")
def _transpose(ex, s, node, x):
    perm = node.attrs.get("perm") or list(reversed(range(len(x.shape))))
    return Spec([x.shape[p] for p in perm], x.dtype)


@rule("Concat")
def _concat(ex, s, node, *xs):
    ax = int(node.attrs["axis"])
    shapes = [list(x.shape) for x in xs]
    if ax != 0 and s.batch > 1:
        shapes = [[s.batch] + sh[1:] if (sh and sh[0] == 1) else sh for sh in shapes]
    ax = ax + len(shapes[0]) if ax < 0 else ax
    out = list(shapes[0])
    out[ax] = sum(sh[ax] for sh in shapes)
    dt = xs[0].dtype
    for x in xs[1:]:
        dt = _dtype2(Spec((), dt), x)
    return Spec(out, dt)


@rule("Split")
def _split(ex, s, node, x, split=None):
    ax = int(node.attrs.get("axis", 0))
    ax = ax + len(x.shape) if ax < 0 else ax
    sizes = _values(ex, split, "Split") if split is not None else node.attrs.get("split")
    if sizes is None:
        n = len(node.output)
        sizes = [x.shape[ax] // n] * n
    return [Spec(list(x.shape[:ax]) + [k] + list(x.shape[ax + 1:]), x.dtype) for k in sizes]


@rule("Slice")
def _slice(ex, s, node, x, starts=None, ends=None, axes=None, steps=None):
    starts = _values(ex, starts, "Slice") if starts is not None else node.attrs["starts"]
    ends = _values(ex, ends, "Slice") if ends is not None else node.attrs["ends"]
    axes = _values(ex, axes, "Slice") if axes is not None else node.attrs.get("axes", list(range(len(starts))))
    steps = _values(ex, steps, "Slice") if steps is not None else [1] * len(starts)
    out = list(x.shape)
    for st, en, ax, sp in zip(starts, ends, axes, steps):
        n = x.shape[ax]
        if sp < 0:
            st = max(0, min(n - 1, st + n if st < 0 else st))
            en = max(-1, min(n - 1, en + n if en < 0 else en))
        else:
            st = max(0, min(n, st + n if st < 0 else st))
            en = max(0, min(n, en + n if en < 0 else en))
        out[ax] = len(range(st, en, sp))
    return Spec(out, x.dtype)


@rule("Gather")
def _gather(ex, s, node, x, idx):
    ax = int(node.attrs.get("axis", 0))
    ax = ax + len(x.shape) if ax < 0 else ax
    return Spec(tuple(x.shape[:ax]) + tuple(idx.shape) + tuple(x.shape[ax + 1:]), x.dtype)


def _axes_of(ex, node, extra, what):
    if extra is not None:
        return _values(ex, extra, what)
    return node.attrs.get("axes")


@rule("Squeeze")
def _squeeze(ex, s, node, x, axes=None):
    ax = _axes_of(ex, node, axes, "Squeeze")
    if ax is None:
        return Spec([d for d in x.shape if d != 1], x.dtype)
    ax = {a + len(x.shape) if a < 0 else a for a in ax}
    return Spec([d for i, d in enumerate(x.shape) if not (i in ax and d == 1)], x.dtype)


@rule("Unsqueeze")
def _unsqueeze(ex, s, node, x, axes=None):
    ax = _axes_of(ex, node, axes, "Unsqueeze")
    nd = len(x.shape) + len(ax)
    out = list(x.shape)
    for a in sorted([a + nd if a < 0 else a for a in ax]):
        out.insert(a, 1)
    return Spec(out, x.dtype)


@rule("ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin")
def _reduce(ex, s, node, x, axes=None):
    ax = _axes_of(ex, node, axes, node.op_type)
    keep = bool(node.attrs.get("keepdims", 1))
    nd = len(x.shape)
    ax = set(range(nd)) if ax is None else {a + nd if a < 0 else a for a in ax}
    return Spec([1 if i in ax else d for i, d in enumerate(x.shape) if keep or i not in ax], x.dtype)


@rule("Shape")
def _shape_op(ex, s, node, x):
    return ex._host_ints(torch.tensor(list(x.shape), dtype=torch.int64), list(x.shape))


@rule("ConstantOfShape")
def _cos(ex, s, node, shape):
    v = node.attrs.get("value")
    dt = torch.from_numpy(np.asarray(v)).dtype if v is not None else torch.float32
    return Spec(_values(ex, shape, "ConstantOfShape"), dt)


@rule("Expand")
def _expand(ex, s, node, x, shape):
    return Spec(_bcast(_shape(x), tuple(_values(ex, shape, "Expand"))), x.dtype)


@rule("Pad")
def _pad(ex, s, node, x, pads=None, value=None, axes=None):
    pads = _values(ex, pads, "Pad") if pads is not None else node.attrs["pads"]
    nd = len(x.shape)
    return Spec([x.shape[i] + pads[i] + pads[nd + i] for i in range(nd)], x.dtype)


@rule("Resize", "Upsample")
def _resize(ex, s, node, x, roi=None, scales=None, sizes=None):
    if node.op_type == "Upsample":
        scales = roi
    if sizes is not None and sizes.numel():
        return Spec(tuple(x.shape[:2]) + tuple(_values(ex, sizes, "Resize")[2:]), x.dtype)
    if not _real(scales):
        raise Unsupported("Resize: needs the values of its scales")
    sc = ex._floats(scales)[2:]
    return Spec(tuple(x.shape[:2]) + tuple(int(np.floor(d * f)) for d, f in zip(x.shape[2:], sc)), x.dtype)


class _HostRun:
    """What the executor's ops read off their session, for the constant-folding part of the pass (host tensors)."""
    device = torch.device("cpu")

    def __init__(self, graph, batch):
        self.graph, self.batch = graph, batch

    def match_batch(self, a, b):
        return a, b


def infer(graph, folded, input_names, batch=1):
    """{tensor name: Spec or host tensor} for every node output of `graph` fed `batch` images.

    folded: names of the FakeQuant nodes whose (constant) input the session quantises once at build time — their output is
    a constant of the same shape."""
    from . import executor as ex
    env = {}
    for name, arr in graph.initializer.items():
        a = np.asarray(arr)
        dt = torch.float32 if a.dtype == np.float16 else torch.from_numpy(np.empty(0, a.dtype)).dtype
        if a.size <= SMALL:
            t = torch.from_numpy(np.array(a, order="C"))
            env[name] = ex._host_ints(t.float() if a.dtype == np.float16 else t)
        else:
            env[name] = Spec(a.shape, dt)
    for n in input_names:
        shp = [max(1, int(d)) for d in graph.get_tensor_shape(n)]
        env[n] = Spec([shp[0] * batch] + shp[1:], torch.float32)
    host = _HostRun(graph, batch)
    for node in graph.graph.node:
        if node.name in folded:
            env[node.output[0]] = env[node.input[0]]
            continue
        args = [env[i] if i != "" else None for i in node.input]
        while args and args[-1] is None:
            args.pop()
        if all(a is None or _real(a) for a in args) and node.op_type in ex._OPS and args:
            with torch.no_grad():
                out = ex._OPS[node.op_type](host, node, *args)          # values known: the executor's own op, on the host
        else:
            fn = _RULES.get(node.op_type)
            if fn is None:
                raise Unsupported(f"no shape rule for {node.op_type}")
            out = fn(ex, host, node, *args)
        if isinstance(out, (list, tuple)):
            for o, v in zip(node.output, out):
                env[o] = v
        else:
            env[node.output[0]] = out
    return env


# ------------------------------------------------------------------------------------------ is batching provably safe?
_PER_SAMPLE_UNARY = {"Relu", "LeakyRelu", "Sigmoid", "Tanh", "HardSigmoid", "HardSwish", "Clip", "Gelu", "Erf", "Sqrt", "Exp", "Log",
                     "Abs", "Neg", "Reciprocal", "Floor", "Ceil", "Identity", "Dropout", "Cast"}
_PER_SAMPLE_NCHW = {"Conv", "ConvTranspose", "MaxPool", "AveragePool", "GlobalAveragePool", "GlobalMaxPool", "BatchNormalization",
                    "InstanceNormalization"}
_BINARY = {"Add", "Sub", "Mul", "Div", "Pow", "Eltwise", "PRelu"}


def batch_transparent(graph, folded, input_names, env):
    """True when every node of `graph` provably maps image b of its inputs to image b of its outputs, the images stacked on ONE
    known axis of every tensor (axis 0 of the network inputs; a Transpose / Gather may move it) — so that B images through a
    graph exported for one are B single runs, and the session need not verify that with a batch-2 forward against two batch-1
    forwards (executor.GraphSession.batched_ok: three more forwards at batch sizes the run never uses, 0.14 s of a fresh
    ResNet-50 process).  Conservative: anything that merges, splits or indexes the batch axis (a Reshape beyond a leading
    0 / 1, Gather / Slice / Concat / a reduction on it, an op not listed) answers False and the session verifies dynamically,
    as before.  env: shape_infer.infer's result at batch 1 (ranks and constant values are read from it)."""
    from . import executor as ex
    batch_axis = {}                                     # tensor name -> the axis its images are stacked on
    for n in input_names:
        shp = graph.get_tensor_shape(n)
        if not shp or max(1, int(shp[0])) != 1:
            return False
        batch_axis[n] = 0
    const = set(graph.initializer)

    def rank(name):
        return len(env[name].shape)

    def ax(a, nd):
        return a + nd if a < 0 else a

    def values(name):
        v = env.get(name)
        return ex._ints(v) if _real(v) else None

    for node in graph.graph.node:
        if node.name in folded:
            const.add(node.output[0])
            continue
        ins = [i for i in node.input if i != ""]
        if any(i not in batch_axis and i not in const for i in ins):
            return False
        if all(i in const for i in ins):
            const.update(o for o in node.output if o)
            continue
        op, x = node.op_type, ins[0]
        rest_const = x in batch_axis and all(i in const for i in ins[1:])
        nd = rank(x)
        k = batch_axis.get(x)
        out = None                                       # the outputs' batch axis once the node is proven per-sample
        if op in _PER_SAMPLE_UNARY:
            out = k if rest_const else None
        elif op in _PER_SAMPLE_NCHW:
            out = 0 if rest_const and k == 0 else None
        elif op == "FakeQuant":
            q = graph._qdq.get(node.name)
            if rest_const and q is not None and not (q.per_channel and ax(int(q.axis), nd) == k):
                out = k
        elif op in _BINARY:
            a, b = ins
            if a in batch_axis and b in batch_axis:
                out = batch_axis[a] if (rank(a) == rank(b) and batch_axis[a] == batch_axis[b]) else None
            else:
                t, c = (a, b) if a in batch_axis else (b, a)
                cs, rt, kt = env[c].shape, rank(t), batch_axis[t]
                grow = max(0, len(cs) - rt)             # right-aligned broadcasting: the result has max(rank) axes
                pos = kt + grow - (max(len(cs), rt) - len(cs))
                if pos < 0 or cs[pos] == 1:
                    out = kt + grow
        elif op == "Flatten":
            out = 0 if rest_const and k == 0 and ax(int(node.attrs.get("axis", 1)), nd) >= 1 else None
        elif op == "Gemm":
            out = 0 if rest_const and k == 0 and not node.attrs.get("transA", 0) and nd == 2 else None
        elif op == "MatMul":
            a, b = ins
            if a in batch_axis and b in batch_axis:
                if rank(a) == rank(b) >= 3 and batch_axis[a] == batch_axis[b] < rank(a) - 2:
                    out = batch_axis[a]
            elif a in batch_axis and nd >= 2 and rank(b) == 2 and k <= nd - 2:
                out = k
        elif op == "Softmax":
            out = k if rest_const and ax(int(node.attrs.get("axis", -1)), nd) != k else None
        elif op == "LayerNormalization":
            out = k if rest_const and k < ax(int(node.attrs.get("axis", -1)), nd) else None
        elif op in ("ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin", "Squeeze"):
            axes = node.attrs.get("axes") if len(ins) < 2 else values(ins[1])
            if rest_const and axes is not None:
                axes = [ax(int(a), nd) for a in axes]
                if k not in axes:
                    gone = op == "Squeeze" or not node.attrs.get("keepdims", 1)
                    out = k - (sum(1 for a in axes if a < k) if gone else 0)
        elif op == "Unsqueeze":
            axes = node.attrs.get("axes") if len(ins) < 2 else values(ins[1])
            if rest_const and axes is not None:
                new = sorted(ax(int(a), nd + len(axes)) for a in axes)
                old = [i for i in range(nd + len(axes)) if i not in new]
                out = old[k]
        elif op == "Concat":
            ca = ax(int(node.attrs["axis"]), nd)
            ks = {batch_axis[i] for i in ins if i in batch_axis}
            cs_ok = all(i in batch_axis or (len(env[i].shape) == nd and env[i].shape[0] == 1) for i in ins)
            if len(ks) == 1 and cs_ok and ca not in ks and all(rank(i) == nd for i in ins):
                kk = next(iter(ks))
                if kk == 0 or all(i in batch_axis for i in ins):    # (the executor expands constants along axis 0 only)
                    out = kk
        elif op == "Transpose":
            perm = node.attrs.get("perm") or list(reversed(range(nd)))
            out = list(perm).index(k) if rest_const else None
        elif op == "Reshape":
            shp = values(ins[1]) if len(ins) > 1 else None
            if rest_const and k == 0 and shp and shp[0] in (0, 1) and env[x].shape[0] == 1 and shp.count(-1) <= 1:
                out = 0
        elif op == "Split":
            out = k if rest_const and ax(int(node.attrs.get("axis", 0)), nd) != k else None
        elif op == "Gather":
            ga = ax(int(node.attrs.get("axis", 0)), nd)
            if rest_const and ga != k:
                out = k + (len(env[ins[1]].shape) - 1 if ga < k else 0)
        elif op == "Slice":
            axes = node.attrs.get("axes") if len(ins) < 4 else values(ins[3])
            if rest_const and axes is not None and all(ax(int(a), nd) != k for a in axes):
                out = k
        elif op == "Pad":
            pads = node.attrs.get("pads") if len(ins) < 2 else values(ins[1])
            if rest_const and pads is not None and len(pads) == 2 * nd and pads[k] == 0 and pads[nd + k] == 0:
                out = k
        if out is None:
            return False
        for o in node.output:
            if o:
                batch_axis[o] = out
    return True
