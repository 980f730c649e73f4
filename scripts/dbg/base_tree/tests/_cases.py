"""Seeded calibration-activation generators shared by the golden-vector generator
(tests/golden/gen_golden.py, which runs the *reference* on them in the build container)
and by the parity tests (which run the oracle and the HIP path on the same data).

Nothing here touches /root/reference.  Determinism relies on numpy's PCG64 stream, which is
identical here and on the GPU box (same image, numpy 2.2.x); every fixture additionally stores
a checksum of the regenerated data so a stream change would be detected rather than mis-compared.
"""
import zlib

import numpy as np

KINDS = ("normal", "relu", "laplace", "uniform", "zeros", "spike", "edges", "neg_only", "tiny", "with_nan")
SIZES = (1000, 2048, 25088, 150528, 802816)


def make_tensor(kind, n, seed):
    """One fp32 activation tensor of `n` elements, flat.  `kind` picks the distribution."""
    rng = np.random.default_rng([0xD1900, seed, n])
    if kind == "normal":
        x = rng.standard_normal(n, dtype=np.float32) * np.float32(1.0 + 0.1 * (seed % 17))
    elif kind == "relu":  # ~50 % exact zeros: worst case for bin-0 contention
        x = np.maximum(rng.standard_normal(n, dtype=np.float32) * np.float32(2.5), np.float32(0))
    elif kind == "laplace":
        x = rng.laplace(0.0, 0.7, n).astype(np.float32)
    elif kind == "uniform":
        x = rng.uniform(-3.0, 5.0, n).astype(np.float32)
    elif kind == "zeros":
        x = np.zeros(n, np.float32)
    elif kind == "spike":  # one huge outlier, the rest tiny: nearly everything lands in bin 0
        x = rng.standard_normal(n, dtype=np.float32) * np.float32(1e-3)
        x[n // 3] = np.float32(1234.5)
    elif kind == "edges":  # values exactly on fp32 bin edges and their fp32 neighbours
        dmax = np.float32(7.3125 + seed)
        bins = 2048 if seed % 2 == 0 else 1000
        step = np.float32(dmax / np.float32(bins))
        i = rng.integers(0, bins + 1, n).astype(np.float32)
        e = i * step
        e[i == bins] = dmax
        jitter = rng.integers(-1, 2, n)
        x = np.where(jitter < 0, np.nextafter(e, np.float32(-np.inf)),
                     np.where(jitter > 0, np.nextafter(e, np.float32(np.inf)), e)).astype(np.float32)
        x = np.clip(x, 0, dmax)
        x[0] = dmax  # make sure the range maximum itself is present
        x = x * rng.choice(np.array([-1, 1], np.float32), n)
    elif kind == "neg_only":
        x = -np.abs(rng.standard_normal(n, dtype=np.float32)) - np.float32(0.25)
    elif kind == "tiny":  # data_min within 1e-6 of zero (dynamic_sym trigger), small magnitudes
        x = np.abs(rng.standard_normal(n, dtype=np.float32)) * np.float32(1e-2)
        x[1] = np.float32(3e-7)
    elif kind == "with_nan":  # numpy max/min propagate it; np.histogram then refuses the range
        x = rng.standard_normal(n, dtype=np.float32)
        x[rng.integers(0, n, 3)] = np.float32(np.nan)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x, dtype=np.float32)


def checksum(x):
    return int(zlib.crc32(np.ascontiguousarray(x).view(np.uint8)))


# ---- a miniature "network": names, sizes and distributions of the tensors one image yields ----
MINI_NET = (
    # name, elements, kind
    ("input", 3 * 32 * 32, "normal"),
    ("conv1", 8192, "normal"),
    ("relu1", 8192, "relu"),
    ("pool1", 2048, "relu"),
    ("fc", 1000, "laplace"),
    ("dead", 512, "zeros"),
)


def mini_net_activations(image_idx):
    """OrderedDict-like list [(name, fp32 array)] for calibration image `image_idx`."""
    out = []
    for t, (name, n, kind) in enumerate(MINI_NET):
        x = make_tensor(kind, n, 1000 * t + image_idx)
        if name == "dead" and image_idx == 5:
            # one image wakes the dead tensor up so its OCTAV mean mixes NaN and finite values
            x = make_tensor("relu", n, 77)
        out.append((name, x))
    return out


# ------------------------------------------------------------------------------------------------ auxiliary fixtures
# (tests/golden/gen_golden_aux.py records what the REFERENCE computes for these; the tests rebuild the same inputs)
def aux_cos_pair(i):
    """Tensor pairs for cos_similarity (utils.py:273-278), incl. an exactly-zero dot product and an all-zero tensor."""
    rng = np.random.default_rng(900 + i)
    if i == 0:
        a = rng.standard_normal(1000).astype(np.float32)
        return a, (a + rng.standard_normal(1000).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    if i == 1:
        a = rng.standard_normal((3, 4, 5)).astype(np.float32)
        return a, (a * np.float32(0.9) + np.float32(0.01)).astype(np.float32)
    if i == 2:
        return np.array([1.0, 0.0, 2.0, 0.0], np.float32), np.array([0.0, 3.0, 0.0, -1.0], np.float32)   # dot == 0
    if i == 3:
        return np.zeros(64, np.float32), rng.standard_normal(64).astype(np.float32)
    if i == 4:
        a = rng.standard_normal(4096).astype(np.float32)
        return a, a.copy()
    if i == 5:
        a = np.maximum(rng.standard_normal(150528), 0).astype(np.float32) * np.float32(3.0)
        return a, (np.round(a / np.float32(0.05)) * np.float32(0.05)).astype(np.float32)
    a = rng.standard_normal(2048).astype(np.float32)
    return a, (-a + rng.standard_normal(2048).astype(np.float32) * np.float32(0.3)).astype(np.float32)


def aux_stack(i, n, C, hw):
    """fp / quantised activation stacks of one node over n images, each [1, C, H, W] (Conv) or [1, C] (Gemm)."""
    rng = np.random.default_rng(700 + i)
    shape = (n, 1, C) + (tuple(hw) if hw else ())
    fp = rng.standard_normal(shape).astype(np.float32) * np.float32(2.0)
    q = (fp + rng.standard_normal(shape).astype(np.float32) * np.float32(0.03) + np.float32(0.01)).astype(np.float32)
    return fp, q


# A 13-node graph that exercises every selection rule of quantize.py:20-108: merge-ReLU behind Conv / Add, a tensor
# feeding two quantised nodes (dedupe), TensorRT's first-Conv-branch-of-an-Add rule, ConvTranspose weights, a two-input
# Mul (RELU_TYPE but not merged), a PRelu fed by the network input (skipped), bias inputs, two network outputs.
AUX_GRAPH = {
    "inputs": ["data"],
    "outputs": ["prob", "pr"],
    "initializers": {"w1": 4, "b1": 4, "w2": 4, "w3": 4, "b3": 4, "wt": 4, "wf": 3, "bf": 3, "slope": 1},
    "tensors": ["data", "c1", "r1", "p1", "c2", "c3", "a1", "r2", "d1", "m1", "g1", "logits", "prob", "pr"],
    "nodes": [
        {"name": "conv1", "op": "Conv", "in": ["data", "w1", "b1"], "out": ["c1"]},
        {"name": "relu1", "op": "Relu", "in": ["c1"], "out": ["r1"]},
        {"name": "pool", "op": "MaxPool", "in": ["r1"], "out": ["p1"]},
        {"name": "conv2", "op": "Conv", "in": ["p1", "w2"], "out": ["c2"]},
        {"name": "conv3", "op": "Conv", "in": ["p1", "w3", "b3"], "out": ["c3"]},
        {"name": "add", "op": "Add", "in": ["c2", "c3"], "out": ["a1"]},
        {"name": "relu_b", "op": "Relu", "in": ["a1"], "out": ["r2"]},
        {"name": "deconv", "op": "ConvTranspose", "in": ["r2", "wt"], "out": ["d1"]},
        {"name": "mul", "op": "Mul", "in": ["d1", "r2"], "out": ["m1"]},
        {"name": "gap", "op": "AveragePool", "in": ["m1"], "out": ["g1"]},
        {"name": "fc", "op": "Gemm", "in": ["g1", "wf", "bf"], "out": ["logits"]},
        {"name": "sig", "op": "Sigmoid", "in": ["logits"], "out": ["prob"]},
        {"name": "prelu", "op": "PRelu", "in": ["data", "slope"], "out": ["pr"]},
    ],
}
