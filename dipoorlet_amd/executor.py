"""Graph executor on PyTorch-ROCm — replaces `ort.InferenceSession(model-with-every-node-output-exposed)`
(dipoorlet/forward_net.py:193-202, 216).

The dense math (conv / GEMM / pooling) is library work through torch (MIOpen / hipBLASLt) and is not part
of the hand-written-kernel scope; what matters for calibration is that EVERY node output of a batch of
B images stays in HBM and is handed to the statistics kernels as-is.  FakeQuant nodes (the fused
QuantizeLinear -> DequantizeLinear pair, quantize.QDQNode) run on k_fake_quant.

Batching: the reference feeds one image per forward (model input batch dim 1).  Here B images are
stacked on dim 0; Reshape targets and dim-0 broadcasts that were constant-folded for batch 1 are
re-scaled to B.
"""
import atexit
import os

import numpy as np
import torch
import torch.nn.functional as F

# Calibration values must come from fp32 arithmetic: no reduced-precision paths in the library kernels of the forward
# (convolutions default to allow_tf32 = True in PyTorch)
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
# DPL_DETERMINISTIC=1: the library's deterministic algorithms in every process that executes graphs (the CLI, its workers, tests).
# MIOpen's default for a 3 x 3 stride-2 convolution on gfx950 — igemm_fwd_gtcx35_nhwc_fp32_*, split over the reduction with fp32
# atomic adds into a zeroed output — differs by 1e-6 from call to call; with this set the Winograd kernel runs instead (181 us
# against 135 us at batch 32, three convolutions of a ResNet-50 forward) and every convolution repeats bit for bit (DESIGN.md
# section 4; scripts/conv_repro_probe.py)
if os.environ.get("DPL_DETERMINISTIC") == "1":
    torch.backends.cudnn.deterministic = True
# MIOpen's find mode, unless the caller has chosen one.  The library's default (DYNAMIC_HYBRID) answers a convolution configuration
# its find-db does not hold by BENCHMARKING every applicable solver — 8 launches of its naive kernel among them — and remembers the
# result in the user find-db (~/.config/miopen): on a fresh account or container that is 0.7 - 4.8 s of a process' first forward
# (ResNet-50 at batch 16 / 64; ResNet-18 3.8 s, ViT-B/16 0.2 s; profiles/r06/conv_repro.md section 3, scripts/find_mode_probe.sh),
# where a calibration run over 1 024 images takes 0.35 s.  FAST takes the library's heuristic choice on a miss and benchmarks
# nothing: 0.06 - 0.07 s on the same cold boxes, the steady forward within 0 - 3 % (the convolutions' own GPU time + 1 - 5 %), a
# process on a warm find-db unchanged.  DPL_MIOPEN_FIND_MODE=library leaves the library's default; any other value is passed on.
_find_mode = os.environ.get("DPL_MIOPEN_FIND_MODE", "FAST")
if _find_mode != "library":
    os.environ.setdefault("MIOPEN_FIND_MODE", _find_mode)

from .forward_net import ActivationSession
from .forward_net import wall as _wall
from .utils import logger, mark

_OPS = {}

_WARM = {}     # "kernels" / "blas": the helper threads (True once joined); "context_s", "kernels_s", "blas_s": their own clocks


def warm_libraries(device=None, blas=True):
    """Starts (once per process) two helper threads that make the FIRST calls of the libraries a forward uses, on tiny tensors
    and streams of their own, while the main thread reads the model, packs the initializers and starts the .bin reader (torch
    ops release the GIL; measured on MI355X, scripts/warm_probe.py: made one after the other the first calls take 0.39 s after
    hipInit, from two threads 0.26 s):
      'blas'     the first GEMM — hipBLASLt (rocBLAS alike) loads its kernel library for gfx950 then: 0.17 - 0.2 s.  Only an op that
                 multiplies matrices THROUGH THE LIBRARY waits for it (wait_warm('blas') in MatMul / Gemm).  While it loads, every
                 other code object the first forward needs queues behind it (scripts/e2e_blas_ab.sh: the first forward is issued
                 70 - 80 ms later), so the CLI passes blas=False here and starts this thread (warm_blas) only for a graph that
                 needs the library: a convolutional network's classifier head runs on ops.gemm_small instead
      'kernels'  the first launch of each family of torch's own kernels (code objects are loaded lazily: 10 - 25 ms each) and
                 MIOpen's first convolution; nothing waits for this thread (first calls are serialised by the libraries' own
                 locks: a forward that gets there first pays the first call itself)
    A fresh ResNet-50 process: first forward 248 ms -> 52 - 92 ms (scripts/startup_probe.py)."""
    if not torch.cuda.is_available():
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    todo = []

    def run(name, body):
        import time
        t0 = time.perf_counter()
        mark(f"warm:{name}:start")
        try:
            torch.cuda.set_device(dev)
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream(dev)):
                body(t0)
                torch.cuda.current_stream(dev).synchronize()
        except Exception:   # noqa: BLE001  (best effort: whatever did not get warm is paid by the first forward, as before)
            pass
        _WARM[name + "_s"] = time.perf_counter() - t0
        mark(f"warm:{name}:end")

    def blas_body(t0):
        a = torch.zeros(8, 64, device=dev)
        torch.addmm(torch.zeros(64, device=dev), a, torch.zeros(64, 64, device=dev))
        torch.matmul(torch.zeros(2, 4, 8, 8, device=dev), torch.zeros(2, 4, 8, 8, device=dev))

    def kernels(t0):
        import time
        x = torch.zeros(2, 8, 16, 16, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        _WARM["context_s"] = time.perf_counter() - t0       # (the HIP context and the first code object, if this thread gets there first)
        y = F.conv2d(x, torch.zeros(8, 8, 3, 3, device=dev), torch.zeros(8, device=dev), 1, 1)
        y = F.max_pool2d(torch.relu(y), 3, 2, 1)
        y = y + y
        y.mean((2, 3), keepdim=True)
        y.abs().amax()
        y.transpose(0, 1).contiguous()
        torch.softmax(y, -1)
        torch.erf(y) * y
        F.layer_norm(y, y.shape[-1:])
        torch.cat([y, y], 1)

    import threading
    if blas and "blas" not in _WARM:
        todo.append(("blas", blas_body))
    if "kernels" not in _WARM:
        todo.append(("kernels", kernels))
    for name, body in todo:
        _WARM[name] = threading.Thread(target=run, args=(name, body), daemon=True, name="dpl-warm-" + name)
        _HELPERS.append(_WARM[name])
        _WARM[name].start()


def warm_blas(device=None):
    """The 'blas' helper thread of warm_libraries alone (no-op if it has been started): for a graph whose matrix products go to
    the library."""
    warm_libraries(device, blas=True)


_HELPERS = []      # every helper thread this module has started (warm-up, convolution pre-warm)


def join_helpers(timeout=10.0):
    """Waits for the helper threads (they run for a few tenths of a second): a process that leaves while one of them is
    inside a library's initialisation ends in that library's static destructors instead of with its exit code."""
    for t in list(_HELPERS):
        if t.is_alive():
            t.join(timeout)
    _HELPERS.clear()


atexit.register(join_helpers)     # (any process that started helper threads — a script, a test — leaves only after them)


def wait_warm(name):
    t = _WARM.get(name)
    if t is not None and t is not True:
        with _wall(f"warm_wait_{name}_s"):
            t.join()
        _WARM[name] = True


def op(*names):
    def deco(fn):
        for n in names:
            _OPS[n] = fn
        return fn
    return deco


def _ints(v):
    if isinstance(v, torch.Tensor):
        known = getattr(v, "_dpl_ints", None)   # integer constants carry their host values (_host_ints): no device read-back
        if known is not None:
            return list(known)
        return [int(x) for x in v.reshape(-1).tolist()]
    return [int(x) for x in np.asarray(v).reshape(-1).tolist()]


def _host_ints(t, src=None):
    """Tag a small constant with its values as Python numbers — integers (Reshape shapes, Gather indices, Slice bounds: the
    shape arithmetic of exported graphs) and floating-point scalars (Clip bounds, Pad value, Resize scales) — so that the
    ops reading them (_ints / _floats) do not synchronise the host with the device on every forward.  src: where to read
    the values (a host copy or a list; default: t itself, one read-back now)."""
    if not isinstance(t, torch.Tensor) or t.dtype == torch.bool:
        return t
    if t.is_floating_point():
        if t.numel() <= 16:
            t._dpl_floats = [float(x) for x in (src if isinstance(src, list) else (t if src is None else src).reshape(-1).tolist())]
    elif t.numel() <= 4096:
        t._dpl_ints = [int(x) for x in (src if isinstance(src, list) else (t if src is None else src).reshape(-1).tolist())]
    return t


def _floats(v):
    known = getattr(v, "_dpl_floats", None)
    return list(known) if known is not None else [float(x) for x in v.reshape(-1).tolist()]


def _pads_nd(pads, nd):
    """ONNX [b1..bn, e1..en] -> (symmetric tuple or None, F.pad list)."""
    pads = list(pads) if pads else [0] * (2 * nd)
    beg, end = pads[:nd], pads[nd:]
    sym = tuple(beg) if beg == end else None
    fpad = []
    for b, e in zip(reversed(beg), reversed(end)):
        fpad += [b, e]
    return sym, fpad


def _auto_pad(node, x, kernel, strides, dilations):
    ap = node.attrs.get("auto_pad", "NOTSET")
    nd = len(kernel)
    if ap in ("NOTSET", ""):
        return node.attrs.get("pads", [0] * (2 * nd))
    if ap == "VALID":
        return [0] * (2 * nd)
    beg, end = [], []
    for i in range(nd):
        size = x.shape[2 + i]
        out = -(-size // strides[i])
        total = max(0, (out - 1) * strides[i] + (kernel[i] - 1) * dilations[i] + 1 - size)
        lo = total // 2 if ap == "SAME_UPPER" else total - total // 2
        beg.append(lo)
        end.append(total - lo)
    return beg + end


@op("Conv")
def _conv(s, node, x, w, b=None):
    nd = w.dim() - 2
    strides = node.attrs.get("strides", [1] * nd)
    dil = node.attrs.get("dilations", [1] * nd)
    pads = _auto_pad(node, x, list(w.shape[2:]), strides, dil)
    sym, fpad = _pads_nd(pads, nd)
    if sym is None:
        x = F.pad(x, fpad)
        sym = (0,) * nd
    fn = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[nd]
    return fn(x, w, b, tuple(strides), sym, tuple(dil), int(node.attrs.get("group", 1)))


@op("ConvTranspose")
def _convt(s, node, x, w, b=None):
    nd = w.dim() - 2
    strides = node.attrs.get("strides", [1] * nd)
    dil = node.attrs.get("dilations", [1] * nd)
    sym, _ = _pads_nd(node.attrs.get("pads"), nd)
    if sym is None:
        raise NotImplementedError("ConvTranspose with asymmetric pads")
    fn = {1: F.conv_transpose1d, 2: F.conv_transpose2d, 3: F.conv_transpose3d}[nd]
    return fn(x, w, b, tuple(strides), sym, tuple(node.attrs.get("output_padding", [0] * nd)),
              int(node.attrs.get("group", 1)), tuple(dil))


@op("Relu")
def _relu(s, node, x):
    return torch.relu(x)


@op("LeakyRelu")
def _lrelu(s, node, x):
    return F.leaky_relu(x, float(node.attrs.get("alpha", 0.01)))


@op("PRelu")
def _prelu(s, node, x, slope):
    return torch.where(x >= 0, x, x * slope)


@op("Sigmoid")
def _sigmoid(s, node, x):
    return torch.sigmoid(x)


@op("Tanh")
def _tanh(s, node, x):
    return torch.tanh(x)


@op("HardSigmoid")
def _hsig(s, node, x):
    return torch.clamp(x * float(node.attrs.get("alpha", 0.2)) + float(node.attrs.get("beta", 0.5)), 0, 1)


@op("HardSwish")
def _hswish(s, node, x):
    return F.hardswish(x)


@op("Clip")
def _clip(s, node, x, lo=None, hi=None):
    lo = node.attrs.get("min") if lo is None else _floats(lo)[0]
    hi = node.attrs.get("max") if hi is None else _floats(hi)[0]
    return torch.clamp(x, min=lo, max=hi)


@op("Gelu")
def _gelu(s, node, x):
    return F.gelu(x, approximate="tanh" if node.attrs.get("approximate", "none") == "tanh" else "none")


for _n, _f in (("Erf", torch.erf), ("Sqrt", torch.sqrt), ("Exp", torch.exp), ("Log", torch.log), ("Abs", torch.abs),
               ("Neg", torch.neg), ("Reciprocal", torch.reciprocal), ("Floor", torch.floor), ("Ceil", torch.ceil)):
    _OPS[_n] = (lambda f: (lambda s, node, x: f(x)))(_f)

for _n, _f in (("Add", torch.add), ("Sub", torch.sub), ("Mul", torch.mul), ("Div", torch.div), ("Pow", torch.pow)):
    _OPS[_n] = (lambda f: (lambda s, node, a, b: f(*s.match_batch(a, b))))(_f)
_OPS["Eltwise"] = _OPS["Add"]


@op("Softmax")
def _softmax(s, node, x):
    return torch.softmax(x, int(node.attrs.get("axis", -1)))


def _pool_args(node, x):
    k = node.attrs["kernel_shape"]
    nd = len(k)
    strides = node.attrs.get("strides", [1] * nd)
    dil = node.attrs.get("dilations", [1] * nd)
    pads = _auto_pad(node, x, k, strides, dil)
    return k, nd, strides, dil, pads


@op("MaxPool")
def _maxpool(s, node, x):
    k, nd, strides, dil, pads = _pool_args(node, x)
    sym, fpad = _pads_nd(pads, nd)
    if sym is None:
        x = F.pad(x, fpad, value=float("-inf"))
        sym = (0,) * nd
    fn = {1: F.max_pool1d, 2: F.max_pool2d, 3: F.max_pool3d}[nd]
    return fn(x, tuple(k), tuple(strides), sym, tuple(dil), bool(node.attrs.get("ceil_mode", 0)))


@op("AveragePool")
def _avgpool(s, node, x):
    k, nd, strides, dil, pads = _pool_args(node, x)
    sym, fpad = _pads_nd(pads, nd)
    cip = bool(node.attrs.get("count_include_pad", 0))
    if sym is None:
        if not cip:
            raise NotImplementedError("AveragePool: asymmetric pads with count_include_pad=0")
        x = F.pad(x, fpad)
        sym = (0,) * nd
    fn = {1: F.avg_pool1d, 2: F.avg_pool2d, 3: F.avg_pool3d}[nd]
    return fn(x, tuple(k), tuple(strides), sym, bool(node.attrs.get("ceil_mode", 0)), cip)


@op("GlobalAveragePool")
def _gap(s, node, x):
    return x.mean(tuple(range(2, x.dim())), keepdim=True)


@op("GlobalMaxPool")
def _gmp(s, node, x):
    return x.amax(tuple(range(2, x.dim())), keepdim=True)


def small_gemm(m, n, k, *tensors):
    """Does a product of these sizes run on ops.gemm_small (the library's own kernel: torch's BLAS path — hipBLASLt — is not initialised for it) rather than
    hipBLASLt?  The classifier head of a convolutional network does; a transformer's products do not, nor does a product that
    autograd has to see (AdaRound / BRECQ reconstruct a Gemm layer through this executor).  DPL_GEMM_SMALL=0: never."""
    from .ops import GEMM_SMALL_MAX
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    return m * n * max(k, 1) <= GEMM_SMALL_MAX and os.environ.get("DPL_GEMM_SMALL", "1") != "0"


@op("MatMul")
def _matmul(s, node, a, b):
    if a.is_cuda and b.dim() == 2 and a.dim() >= 2 and a.dtype == torch.float32 and b.dtype == torch.float32 \
            and small_gemm(a.numel() // a.shape[-1], b.shape[1], a.shape[-1], a, b):
        from . import ops
        return ops.gemm_small(a.reshape(-1, a.shape[-1]), b).reshape(*a.shape[:-1], b.shape[1])
    if a.is_cuda:
        wait_warm("blas")
    return torch.matmul(a, b)


@op("Gemm")
def _gemm(s, node, a, b, c=None):
    if node.attrs.get("transA", 0):
        a = a.t()
    if node.attrs.get("transB", 0):
        b = b.t()
    alpha, beta = float(node.attrs.get("alpha", 1.0)), float(node.attrs.get("beta", 1.0))
    if a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == torch.float32 and b.dtype == torch.float32 \
            and (c is None or (c.dim() <= 2 and c.dtype == torch.float32)) and small_gemm(a.shape[0], b.shape[1], a.shape[1], a, b, c):
        from . import ops
        return ops.gemm_small(a, b, c, alpha, beta)       # (b: a transposed VIEW for transB = 1 — the kernel takes strides)
    if a.is_cuda:
        wait_warm("blas")
    if c is not None and a.dim() == 2 and b.dim() == 2 and c.dim() <= 2:
        return torch.addmm(c, a, b, beta=beta, alpha=alpha)      # one hipBLASLt call, bias in the epilogue
    y = torch.matmul(a, b)
    if alpha != 1.0:
        y = y * alpha
    if c is not None:
        y = y + (c if beta == 1.0 else c * beta)
    return y


@op("Flatten")
def _flatten(s, node, x):
    ax = int(node.attrs.get("axis", 1))
    ax = ax + x.dim() if ax < 0 else ax
    lead = int(np.prod(x.shape[:ax])) if ax > 0 else 1
    return x.reshape(lead, -1)


@op("Reshape")
def _reshape(s, node, x, shape):
    shp = _ints(shape)
    if not node.attrs.get("allowzero", 0):
        shp = [x.shape[i] if d == 0 else d for i, d in enumerate(shp)]
    # the graph was exported (and constant-folded) for batch 1: re-scale a literal leading 1 to the live batch
    if s.batch > 1 and shp and shp[0] == 1 and x.shape[0] == s.batch:
        want = int(np.prod([d for d in shp if d > 0]))
        if (-1 in shp and x.numel() % (want * s.batch) == 0) or (-1 not in shp and want * s.batch == x.numel()):
            shp[0] = s.batch
    return x.reshape(shp)


@op("Transpose")
def _transpose(s, node, x):
    perm = node.attrs.get("perm") or list(reversed(range(x.dim())))
    return x.permute(perm)


@op("Concat")
def _concat(s, node, *xs):
    ax = int(node.attrs["axis"])
    if ax != 0 and s.batch > 1:
        xs = [x.expand(s.batch, *x.shape[1:]) if (x.dim() > 0 and x.shape[0] == 1) else x for x in xs]
    return torch.cat(list(xs), ax)


@op("Split")
def _split(s, node, x, split=None):
    ax = int(node.attrs.get("axis", 0))
    sizes = _ints(split) if split is not None else node.attrs.get("split")
    if sizes is None:
        n = len(node.output)
        sizes = [x.shape[ax] // n] * n
    return list(torch.split(x, sizes, ax))


@op("Slice")
def _slice(s, node, x, starts=None, ends=None, axes=None, steps=None):
    starts = _ints(starts) if starts is not None else node.attrs["starts"]
    ends = _ints(ends) if ends is not None else node.attrs["ends"]
    axes = _ints(axes) if axes is not None else node.attrs.get("axes", list(range(len(starts))))
    steps = _ints(steps) if steps is not None else [1] * len(starts)
    idx = [slice(None)] * x.dim()
    for st, en, ax, sp in zip(starts, ends, axes, steps):
        n = x.shape[ax]
        if sp < 0:      # (the exporter reverses the pads vector of F.pad this way) start in [0, n-1], end in [-1, n-1]
            st = max(0, min(n - 1, st + n if st < 0 else st))
            en = max(-1, min(n - 1, en + n if en < 0 else en))
            # (an empty range — start <= end with a negative step — is an empty result in ONNX; torch.arange refuses it)
            x = torch.index_select(x, ax, torch.tensor(list(range(st, en, sp)), dtype=torch.long, device=x.device))
            continue
        st = max(0, min(n, st + n if st < 0 else st))
        en = max(0, min(n, en + n if en < 0 else en))
        idx[ax] = slice(st, en, sp)
    return x[tuple(idx)]


@op("Gather")
def _gather(s, node, x, idx):
    ax = int(node.attrs.get("axis", 0))
    if idx.dim() == 0:
        return x.select(ax, _ints(idx)[0])
    idx = idx.long()
    return torch.index_select(x, ax, idx.reshape(-1)).reshape(x.shape[:ax] + tuple(idx.shape) + x.shape[ax + 1:])


def _axes(node, extra):
    if extra is not None:
        return _ints(extra)
    return node.attrs.get("axes")


@op("Squeeze")
def _squeeze(s, node, x, axes=None):
    ax = _axes(node, axes)
    if ax is None:
        return x.squeeze()
    for a in sorted([a + x.dim() if a < 0 else a for a in ax], reverse=True):
        x = x.squeeze(a)
    return x


@op("Unsqueeze")
def _unsqueeze(s, node, x, axes=None):
    ax = _axes(node, axes)
    nd = x.dim() + len(ax)
    for a in sorted([a + nd if a < 0 else a for a in ax]):
        x = x.unsqueeze(a)
    return x


def _reduce(fn):
    def run(s, node, x, axes=None):
        ax = _axes(node, axes)
        keep = bool(node.attrs.get("keepdims", 1))
        if ax is None:
            ax = list(range(x.dim()))
        return fn(x, tuple(ax), keep)
    return run


_OPS["ReduceMean"] = _reduce(lambda x, ax, k: x.mean(ax, keepdim=k))
_OPS["ReduceSum"] = _reduce(lambda x, ax, k: x.sum(ax, keepdim=k))
_OPS["ReduceMax"] = _reduce(lambda x, ax, k: x.amax(ax, keepdim=k))
_OPS["ReduceMin"] = _reduce(lambda x, ax, k: x.amin(ax, keepdim=k))


@op("BatchNormalization")
def _bn(s, node, x, scale, bias, mean, var):
    return F.batch_norm(x, mean, var, scale, bias, False, 0.0, float(node.attrs.get("epsilon", 1e-5)))


@op("LayerNormalization")
def _ln(s, node, x, scale, bias=None):
    ax = int(node.attrs.get("axis", -1))
    ax = ax + x.dim() if ax < 0 else ax
    return F.layer_norm(x, tuple(x.shape[ax:]), scale, bias, float(node.attrs.get("epsilon", 1e-5)))


@op("InstanceNormalization")
def _in(s, node, x, scale, bias):
    return F.instance_norm(x, weight=scale, bias=bias, eps=float(node.attrs.get("epsilon", 1e-5)))


@op("Identity", "Dropout")
def _identity(s, node, x, *rest):
    return x


_CAST = {1: torch.float32, 2: torch.uint8, 3: torch.int8, 6: torch.int32, 7: torch.int64, 9: torch.bool,
         10: torch.float16, 11: torch.float64}


@op("Cast")
def _cast(s, node, x):
    return x.to(_CAST[int(node.attrs["to"])])


@op("Shape")
def _shape(s, node, x):
    return _host_ints(torch.tensor(list(x.shape), dtype=torch.int64, device=x.device), list(x.shape))


@op("ConstantOfShape")
def _cos(s, node, shape):
    v = node.attrs.get("value")
    val = float(np.asarray(v).reshape(-1)[0]) if v is not None else 0.0
    dt = torch.from_numpy(np.asarray(v)).dtype if v is not None else torch.float32
    return torch.full(_ints(shape), val, dtype=dt, device=s.device)


@op("Expand")
def _expand(s, node, x, shape):
    shp = _ints(shape)
    return x.expand(tuple(int(d) for d in np.broadcast_shapes(tuple(x.shape), tuple(shp))))


@op("Where")
def _where(s, node, c, a, b):
    return torch.where(c.bool(), a, b)


@op("Equal")
def _equal(s, node, a, b):
    return a == b


@op("Pad")
def _pad(s, node, x, pads=None, value=None, axes=None):
    pads = _ints(pads) if pads is not None else node.attrs["pads"]
    nd = x.dim()
    _, fpad = _pads_nd(pads, nd)
    mode = node.attrs.get("mode", "constant")
    v = _floats(value)[0] if value is not None else float(node.attrs.get("value", 0.0))
    return F.pad(x, fpad, mode=mode, value=v) if mode == "constant" else F.pad(x, fpad, mode=mode)


@op("Resize", "Upsample")
def _resize(s, node, x, roi=None, scales=None, sizes=None):
    mode = node.attrs.get("mode", "nearest")
    if node.op_type == "Upsample":
        scales = roi
    if sizes is not None and sizes.numel():
        return F.interpolate(x, size=_ints(sizes)[2:], mode={"linear": "bilinear"}.get(mode, mode))
    sc = _floats(scales)[2:]
    return F.interpolate(x, scale_factor=sc, mode={"linear": "bilinear"}.get(mode, mode))


@op("FakeQuant")
def _fake_quant(s, node, x):
    q = s.graph._qdq[node.name]
    if not x.is_cuda:          # shape-inference pass (host or meta tensors): values are irrelevant there
        return x
    return q.apply(x.contiguous())


def fused_fake_quant(s, node, pre, *xs):
    """A FakeQuant node with its producer's ReLU / Add + ReLU applied on the way in (relu_fusion): one k_fake_quant<PRE> launch."""
    q = s.graph._qdq[node.name]
    if pre == "add_relu":
        return q.apply(xs[0].contiguous(), pre=pre, x2=xs[1].contiguous())
    return q.apply(xs[0].contiguous(), pre=pre)


def relu_fusion(graph, folded, consts, keep=(), shape1=None):
    """Which activation FakeQuant nodes of a fake-quantised graph can take their producer's ReLU — or residual Add + ReLU — inside
    the Q/DQ kernel (k_fake_quant<PRE>): the reference's merge-ReLU rule leaves a ReLU behind Conv / Gemm / Add unquantised at its
    input (quantize.py:50-55), so the Q/DQ pair of the next layer's input sits directly behind that ReLU (:74-93).  Separate
    launches move 16 B per element for the pair (ReLU 4 + 4, Q/DQ 4 + 4) and 28 B with the Add in front; fused 8 B / 12 B.

    A ReLU is fused when its output has exactly one consumer — the FakeQuant node —, is no network output and is not asked for by
    name (`keep`: the tensors the caller wants to see; a session that exposes every tensor fuses nothing).  The Add in front of it
    joins under the same conditions when both operands are activations of the output's shape (`shape1`: per-image shapes; no
    broadcast).  DPL_FUSE_RELU=0: never (A/B).

    Returns (fused, skipped): fused[fq_node_name] = (pre, [input names]); skipped = names of the Relu / Add nodes that do not run."""
    if os.environ.get("DPL_FUSE_RELU", "1") == "0":
        return {}, set()
    keep = set(keep) | set(graph.network_outputs)
    nodes = [n for n in graph.graph.node if n.name not in folded]
    uses, producer = {}, {}
    for n in nodes:
        for i in n.input:
            if i != "":
                uses.setdefault(i, []).append(n)
        for o in n.output:
            producer[o] = n
    fused, skipped = {}, set()

    def sole(name, consumer):
        return name not in keep and len(uses.get(name, ())) == 1 and uses[name][0] is consumer

    def shape_of(name):
        """Per-image shape of an activation; a fake-quantised tensor (not listed by a session that does not expose them) has its
        source's."""
        if shape1 is None or name in consts:
            return None
        if shape1.get(name) is not None:
            return shape1[name]
        p = producer.get(name)
        return shape_of(p.input[0]) if p is not None and p.op_type == "FakeQuant" else None

    for q in nodes:
        if q.op_type != "FakeQuant" or q.input[0] in consts:
            continue
        r = producer.get(q.input[0])
        if r is None or r.op_type != "Relu" or len(r.output) != 1 or not sole(r.output[0], q):
            continue
        pre, ins, skip = "relu", [r.input[0]], [r.name]
        a = producer.get(r.input[0])
        if a is not None and a.op_type == "Add" and len(a.input) == 2 and sole(a.output[0], r) and shape_of(a.output[0]) is not None \
                and all(shape_of(i) == shape_of(a.output[0]) for i in a.input):
            pre, ins, skip = "add_relu", list(a.input), [a.name, r.name]
        fused[q.name] = (pre, ins)
        skipped.update(skip)
    return fused, skipped


class GraphSession(ActivationSession):
    """All-outputs session over an ONNXGraph."""

    def __init__(self, graph, device=None, expose_fake_quant=False, first_batch=None):
        """first_batch (optional): a function of the session, called once the per-image shapes are known (before the weights are on
        the device), that returns the number of images the caller's first forward will carry — the session then starts what
        that forward needs at once, beside its own build: the first call of every convolution configuration at that batch size
        (prewarm_convs, on zero weights) and — only if a matrix product of the graph is too large for ops.gemm_small at that
        batch size — the BLAS library's (warm_blas).  Without it the caller may call prewarm_convs itself."""
        self.graph = graph
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.batch = 1
        self.consts = {}
        if self.device.type == "cuda":
            warm_libraries(self.device, blas=first_batch is None)     # (no-op for what the CLI has started already)
        missing = sorted({n.op_type for n in graph.graph.node if n.op_type not in _OPS})
        if missing:
            raise NotImplementedError(f"executor: unsupported ONNX ops {missing}")
        self.expose_fake_quant = expose_fake_quant
        self.input_names = list(graph.network_inputs)
        # fake-quantised WEIGHTS are constants: quantise them once here instead of on every forward
        self._folded = set()
        self._batched_ok = None   # decided by batched_ok() at the first batched run
        fold = [n for n in graph.graph.node if n.op_type == "FakeQuant" and n.input[0] in graph.initializer]
        for node in fold:
            self._folded.add(node.name)
        # shapes first (host rules: no device work), so that the libraries' first calls for THIS graph start before the weights
        # travel; a graph the rules do not cover needs its weights on the device for the batch-1 forward that replaces them
        uploaded = False
        if not self._infer(host_only=True):
            with _wall("session_consts_s"):
                self._upload_consts()
            uploaded = True
            self._fold_weights(fold)
            self._infer(skip_host=True)
        mark("session:shapes_known")
        if first_batch is not None and self.device.type == "cuda":
            nb = int(first_batch(self))
            if self.needs_blas(nb):
                warm_blas(self.device)
            self.prewarm_convs(nb)
            mark("session:conv_threads_started")
        if not uploaded:
            with _wall("session_consts_s"):
                self._upload_consts()
            self._fold_weights(fold)
        mark("session:consts_issued")

    def needs_blas(self, batch):
        """Does a forward of `batch` images multiply matrices through the BLAS library — a MatMul / Gemm that small_gemm()
        refuses (or whose shapes are not both known here)?"""
        for node in self.graph.graph.node:
            if node.op_type not in ("MatMul", "Gemm") or node.name in self._folded:
                continue
            a, b = self._any_shape(node.input[0], batch), self._any_shape(node.input[1], 1)
            if a is None or b is None or len(b) != 2 or len(a) < 2 or (node.op_type == "Gemm" and len(a) != 2):
                return True
            if node.op_type == "Gemm" and node.attrs.get("transA", 0):
                a = a[::-1]
            k = a[-1]
            n = b[0] if (node.op_type == "Gemm" and node.attrs.get("transB", 0)) else b[1]
            if not small_gemm(int(np.prod(a[:-1])), int(n), int(k)):
                return True
        return False

    def _any_shape(self, name, batch):
        """Shape of an initializer, of a folded fake-quantised one, or of an activation at `batch` images; None if unknown."""
        if name in self.graph.initializer:
            return tuple(np.asarray(self.graph.initializer[name]).shape)
        for node in self.graph.graph.node:
            if node.name in self._folded and node.output[0] == name:
                return tuple(np.asarray(self.graph.initializer[node.input[0]]).shape)
        shp = self.shape1.get(name)
        return None if shp is None else (batch * shp[0],) + tuple(shp[1:])

    def _fold_weights(self, nodes):
        """Fake-quantised WEIGHTS are constants: quantised once, here, instead of on every forward — every weight of the graph
        in ONE launch (dpl_fake_quant_items over the set of weight tensors, per-channel rows along each weight's own axis: 54
        tensors for ResNet-50) rather than one launch and two parameter uploads per weight (quantize.py:197-239 builds a Q/DQ
        pair per weight; ONNXRuntime runs each on every inference)."""
        if not nodes:
            return
        ws = [self.consts[n.input[0]] for n in nodes]
        if self.device.type != "cuda" or not all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() > 0 for w in ws):
            for n, w in zip(nodes, ws):
                self.consts[n.output[0]] = _OPS["FakeQuant"](self, n, w) if w.is_cuda else w
            return
        from . import ops
        plan = ops.TensorSetPlan([w.numel() for w in ws], 1, self.device)
        # all scales / zero points of the set in two transfers
        qs = [self.graph._qdq[n.name] for n in nodes]
        sizes = [q.scale.size for q in qs]
        scale = torch.from_numpy(np.concatenate([q.scale for q in qs]).astype(np.float32)).to(self.device)
        zp = torch.from_numpy(np.concatenate([np.broadcast_to(q.zero_point_as_stored(), q.scale.shape) for q in qs]).astype(np.int32)).to(self.device)
        params, off = [], 0
        for q, w, k in zip(qs, ws, sizes):
            lo, hi = q.saturation()
            inner = 1
            if k > 1:       # channel c of element i = (i / inner) % n_channels: inner = the elements behind the channel axis
                if w.shape[q.axis] != k:
                    raise ValueError(f"fake-quant of {q.tensor_name}: {k} channel scales for axis {q.axis} of {tuple(w.shape)}")
                inner = int(np.prod(w.shape[q.axis + 1:])) if q.axis + 1 < w.dim() else 1
            params.append((scale[off:off + k], zp[off:off + k], inner, lo, hi))
            off += k
        outs = ops.FakeQuantSet(plan, params)([w.reshape(1, -1) for w in ws])
        for n, w, y in zip(nodes, ws, outs):
            self.consts[n.output[0]] = y.view(w.shape)

    def _upload_consts(self):
        """Every initializer to the device ONCE: the fp32 ones (weights, biases: 102 MB for ResNet-50) are packed into one
        pinned host buffer and leave in a single asynchronous transfer (161 pageable copies, each a host round trip, took
        a third of a second together with the ranges' second upload); a constant is a view of that one device buffer
        (256-byte aligned).  Integer constants — shapes, indices — keep their host values (_host_ints)."""
        flat, other = [], []
        for name, arr in self.graph.initializer.items():
            a = np.asarray(arr)
            if a.dtype in (np.float32, np.float16) and a.ndim > 0 and a.size > 0:
                flat.append((name, a))
            else:
                other.append((name, a))
        if flat and self.device.type == "cuda":
            offs, total = [], 0
            for _, a in flat:
                offs.append(total)
                total += (a.size + 63) // 64 * 64
            host = torch.empty(total, dtype=torch.float32, pin_memory=True)
            hv = host.numpy()
            for (_, a), o in zip(flat, offs):
                hv[o:o + a.size] = a.reshape(-1)            # (fp16 initializers widen here)
            dev = host.to(self.device, non_blocking=True)
            # pinned source: alive until the copy has run, then released at the session's first forward (a few hundred MB of
            # page-locked memory per session for ViT-B/16; --bc builds two sessions, AdaRound more)
            self._consts_host = host
            self._consts_copied = torch.cuda.Event()
            self._consts_copied.record(torch.cuda.current_stream(self.device))
            for (name, a), o in zip(flat, offs):
                v = dev[o:o + a.size].view(a.shape)
                self.consts[name] = _host_ints(v, [float(x) for x in a.reshape(-1)]) if a.size <= 16 else v
        else:
            other = flat + other
        for name, a in other:
            a = np.array(a, order="C")  # (np.ascontiguousarray would turn a 0-d scalar into shape (1,))
            t = torch.from_numpy(a.astype(np.float32)) if a.dtype == np.float16 else torch.from_numpy(a)
            self.consts[name] = _host_ints(t.to(self.device), t)

    def _infer(self, host_only=False, skip_host=False):
        """(host_only: return False instead of running the device forward when the host rules do not cover the graph; skip_host:
        the rules have been tried.)
        Every tensor's per-image shape (replaces onnx shape inference): on the host, shape_infer's rule per op — no device
        work (a batch-1 forward on zeros cost a fresh process 0.3 s: the libraries load and choose kernels for a batch size
        the run never uses).  A graph with an op that has no rule runs that batch-1 forward instead (DPL_INFER_DEVICE=1
        forces it)."""
        env = None
        if os.environ.get("DPL_INFER_DEVICE", "0") != "1" and not skip_host:
            from . import shape_infer
            try:
                with _wall("session_infer_host_s"):
                    env = shape_infer.infer(self.graph, self._folded, self.input_names, 1)
            except shape_infer.Unsupported as e:
                logger.info("executor: shapes from a batch-1 forward on the device (%s)", e)
                env = None
        if env is None and host_only:
            return False
        if env is None:
            with _wall("session_infer_device_s"):
                feeds = {n: torch.zeros([max(1, int(d)) for d in self.graph.get_tensor_shape(n)], dtype=torch.float32,
                                        device=self.device) for n in self.input_names}
                env = self._forward(feeds, 1)
        if self._batched_ok is None and os.environ.get("DPL_EXECUTOR_VERIFY_BATCHING", "0") != "1":
            from . import shape_infer
            try:    # batching proven per-sample node by node: no batch-2 against batch-1 forwards at the first batched run
                if shape_infer.batch_transparent(self.graph, self._folded, self.input_names, env):
                    self._batched_ok = True
            except Exception:   # noqa: BLE001  (no proof: the session verifies dynamically)
                pass
        names, elems = [], []
        self.shape1 = {}
        for n in self.input_names:
            names.append(n)
            elems.append(env[n].numel())
            self.shape1[n] = tuple(env[n].shape)
        for node in self.graph.graph.node:
            if (node.op_type == "FakeQuant" and not self.expose_fake_quant) or node.name in self._folded:
                continue
            for o in node.output:
                if o == "" or o in names:
                    continue
                t = env[o]
                self.graph.set_tensor_shape(o, list(t.shape))
                if t.is_floating_point():   # integer tensors (Shape, indices) are not calibrated
                    names.append(o)
                    elems.append(t.numel())
                    self.shape1[o] = tuple(t.shape)
        self.tensor_names, self.elems_per_image = names, elems
        return True

    def match_batch(self, a, b):
        return a, b

    def prewarm_convs(self, batch, threads=3):
        """The first call of every DISTINCT convolution configuration of this graph at `batch` images, on zeros, from `threads`
        helper threads (streams of their own; the graph's last layers first, the forward starts at the other end).  MIOpen
        resolves a configuration's solver and loads its code object on first use — 2.7 ms per configuration, 63 ms for
        ResNet-50's 23 from one thread, 18 ms from three, and the main thread's own first forward then finds them loaded
        (scripts/miopen_probe.py).  Returns at once; the session's first forward joins the threads."""
        if self.device.type != "cuda" or batch < 1 or os.environ.get("DPL_PREWARM_CONVS", "1") == "0" \
                or getattr(self, "_prewarmed", False):
            return
        self._prewarmed = True
        seen, todo = set(), []
        for node in self.graph.graph.node:
            if node.op_type != "Conv" or node.name in self._folded:
                continue
            shp, w = self.shape1.get(node.input[0]), self._any_shape(node.input[1], 1)
            if shp is None or w is None or len(shp) != len(w) or shp[0] != 1:
                continue
            key = (tuple(shp), tuple(w), tuple(node.attrs.get("strides", ())), tuple(node.attrs.get("pads", ())),
                   tuple(node.attrs.get("dilations", ())), int(node.attrs.get("group", 1)), node.attrs.get("auto_pad", ""), len(node.input))
            rest = [self._any_shape(i, 1) for i in node.input[1:] if i != ""]
            if key not in seen and all(r is not None for r in rest):
                seen.add(key)
                todo.append((node, (batch,) + tuple(shp[1:]), rest))
        todo.reverse()

        def work(part):
            try:
                torch.cuda.set_device(self.device)
                with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream(self.device)):
                    for node, shape, rest in part:      # (zero weights of the right shapes: the solver is chosen by the configuration)
                        args = [torch.zeros(r, device=self.device) for r in rest]
                        _OPS["Conv"](self, node, torch.zeros(shape, device=self.device), *args)
                    torch.cuda.current_stream(self.device).synchronize()
            except Exception:   # noqa: BLE001  (best effort)
                pass
            mark("warm:convs:end")

        import threading
        self._conv_threads = [threading.Thread(target=work, args=(todo[k::threads],), daemon=True, name=f"dpl-warm-conv{k}")
                              for k in range(max(1, min(threads, len(todo))))]
        for t in self._conv_threads:
            _HELPERS.append(t)
            t.start()

    def set_const(self, name, tensor):
        """Replace an initializer on the device (a weight updated by a weight transform) and refresh the folded
        fake-quantised copy that depends on it."""
        self.consts[name] = _host_ints(tensor.to(self.device))
        for node in self.graph.graph.node:
            if node.name in self._folded and node.input[0] == name:
                self.consts[node.output[0]] = _OPS["FakeQuant"](self, node, self.consts[name])

    def fusion(self, keep):
        """relu_fusion for a forward that hands out the tensors `keep` only (cached per set of names)."""
        key = frozenset(keep)
        cache = self.__dict__.setdefault("_fusion_cache", {})
        if key not in cache:
            cache[key] = relu_fusion(self.graph, self._folded, self.consts, key, getattr(self, "shape1", None))
        return cache[key]

    def _forward(self, feeds, batch, keep=None):
        """keep: the tensors the caller will read (None: any of them — every node runs on its own); a fake-quantised graph then
        runs ReLU -> Q/DQ and Add -> ReLU -> Q/DQ chains as one launch where nothing else reads the tensors in between."""
        self.batch = batch
        env = dict(self.consts)
        env.update(feeds)
        fused, skipped = self.fusion(keep) if keep is not None and self.device.type == "cuda" else ({}, ())
        for node in self.graph.graph.node:
            if node.name in self._folded or node.name in skipped:
                continue
            if node.name in fused:
                pre, ins = fused[node.name]
                env[node.output[0]] = fused_fake_quant(self, node, pre, *[env[i] for i in ins])
                continue
            args = [env[i] if i != "" else None for i in node.input]
            while args and args[-1] is None:
                args.pop()
            out = _OPS[node.op_type](self, node, *args)
            if isinstance(out, (list, tuple)):
                for o, v in zip(node.output, out):
                    env[o] = v
            else:
                env[node.output[0]] = out
        return env

    def _lead(self):
        return max(1, int(self.graph.get_tensor_shape(self.input_names[0])[0]))

    @torch.no_grad()
    def _run_env(self, inputs, batch, keep=None):
        if getattr(self, "_conv_threads", None):
            # the first forward WAITS for the convolution threads instead of racing them (two threads resolving the same
            # configuration at the same time both pay for it): DPL_PREWARM_WAIT=0 races, for A/B
            if os.environ.get("DPL_PREWARM_WAIT", "1") != "0":
                with _wall("warm_wait_convs_s"):
                    for t in self._conv_threads:
                        t.join()
            self._conv_threads = None
        if getattr(self, "_consts_host", None) is not None and self._consts_copied.query():
            self._consts_host = None
        mark("first_forward:start")     # (the other helper threads are not waited for: first calls are serialised by the libraries' own locks)
        return self._forward({n: inputs[n].to(self.device, torch.float32) for n in self.input_names}, batch, keep)

    def _collect(self, env, names, batch):
        out = []
        for n in names:
            t = env[n]
            if t.dtype != torch.float32:
                t = t.float()
            out.append(self._batch_major(n, t, batch) if n in self.shape1 else t)
        return out

    @torch.no_grad()
    def batched_ok(self):
        """Is running B images through a graph exported for one image the same as running them one by one?  The executor
        batches by rewriting literal leading 1s (Reshape, Concat, ...) and by locating the batch axis of every tensor
        (_batch_major): heuristics that a Reshape with a literal non-leading shape can defeat WITHOUT an error — the
        calibration statistics would then silently come from scrambled data.  So, once per session: a batch-2 forward on
        random inputs must reproduce, for every exposed tensor, two batch-1 forwards.  If not, this session runs one image
        at a time from then on (and says so)."""
        if self._batched_ok is None and os.environ.get("DPL_EXECUTOR_PER_IMAGE"):   # (testing aid: the fallback mode)
            self._batched_ok = False
        if self._batched_ok is None:
            import time
            t_check = time.perf_counter()
            g = torch.Generator(device="cpu").manual_seed(20260)
            lead = self._lead()
            feeds = {n: torch.randn([2 * lead] + [max(1, int(d)) for d in self.graph.get_tensor_shape(n)[1:]], generator=g)
                        .to(self.device) for n in self.input_names}
            ok = True
            try:
                both = self._collect(self._run_env(feeds, 2), self.tensor_names, 2)
                for k in range(2):
                    one = self._collect(self._run_env({n: v[k * lead:(k + 1) * lead] for n, v in feeds.items()}, 1),
                                        self.tensor_names, 1)
                    for name, tb, t1 in zip(self.tensor_names, both, one):
                        ref = t1.reshape(-1).double()
                        got = tb[k].reshape(-1).double()
                        # scrambled data is wrong by O(1); different library kernels for the two batch sizes by ~1e-4
                        bad = got.numel() != ref.numel()
                        if not bad:
                            bad = float((got - ref).norm()) > 2e-2 * float(ref.norm()) + 1e-6
                        if bad:
                            logger.warning("executor: batched execution of this graph differs from per-image execution at "
                                           "tensor %s: running one image at a time", name)
                            ok = False
                            break
                    if not ok:
                        break
            except RuntimeError as e:   # e.g. no batch axis could be located
                logger.warning("executor: batched execution of this graph failed (%s): running one image at a time", e)
                ok = False
            self._batched_ok = ok
            from .forward_net import WALL
            WALL["batched_check_s"] = WALL.get("batched_check_s", 0.0) + time.perf_counter() - t_check
        return self._batched_ok

    def _run_any(self, inputs, names, keep=None):
        first = inputs[self.input_names[0]]
        lead = self._lead()
        batch = first.shape[0] // lead
        if batch > 1 and not self.batched_ok():
            per = [self._collect(self._run_env({n: v[k * lead:(k + 1) * lead] for n, v in inputs.items()}, 1, keep), names, 1)
                   for k in range(batch)]
            # [B, per-image ...] like the batched path: images stack on the leading 1 of the per-image shape, or on a new axis
            return [torch.cat([p[i] for p in per]) if (per[0][i].dim() > 0 and per[0][i].shape[0] == 1)
                    else torch.stack([p[i] for p in per]) for i in range(len(names))]
        return self._collect(self._run_env(inputs, batch, keep), names, batch)

    @torch.no_grad()
    def run(self, inputs):
        return self._run_any(inputs, self.tensor_names)

    def _batch_major(self, name, t, batch):
        """Calibration tensors are handed on as [B, per-image...] contiguous.  A graph may carry the batch on
        another axis (e.g. the [3, B, heads, N, d] qkv transpose of an attention block) or not at all."""
        if batch == 1:
            return t.contiguous()
        s1 = self.shape1[name]
        if tuple(t.shape) == s1:                      # batch-independent value: the same for every image
            return t.unsqueeze(0).expand(batch, *t.shape).contiguous()
        diff = [d for d in range(t.dim()) if t.shape[d] != s1[d]]
        if len(s1) != t.dim() or len(diff) != 1 or t.shape[diff[0]] != s1[diff[0]] * batch:
            raise RuntimeError(f"executor: cannot locate the batch axis of {name}: batch-1 shape {s1}, live {tuple(t.shape)}")
        d = diff[0]
        if d == 0:
            return t.contiguous()
        lead = s1[d]
        # split axis d into (batch, lead) and bring batch to the front
        v = t.reshape(*t.shape[:d], batch, lead, *t.shape[d + 1:])
        return v.movedim(d, 0).reshape(batch, *s1).contiguous() if lead == 1 else v.movedim(d, 0).contiguous()

    @torch.no_grad()
    def run_named(self, inputs, names):
        """Chosen tensors by name, laid out like run()'s ([B, per-image ...], checked batching).  Only these are promised to exist:
        a fake-quantised graph fuses the ReLU (and residual Add) in front of a Q/DQ pair into its kernel when their outputs are
        not among `names` (relu_fusion)."""
        names = list(names)
        return self._run_any(inputs, names, keep=names)
