"""CPU ORACLE (test infrastructure, NOT product code) — numpy restatement of the reference's
activation-calibration arithmetic.

  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
    The product package (dipoorlet_amd/) never does; its ops fail loudly without the HIP library.
  * Every function cites the reference file:line (relative to /root/reference) it restates.
  * Parity pin: tests/test_oracle_golden.py checks every function here against golden vectors that
    tests/golden/gen_golden.py produced by running the reference's own code (numpy 2.2.6) in the
    build container.  np.histogram itself is third-party (numpy, unpinned in requirements.txt:5;
    2.2.6 installed) — `abs_hist` restates its uniform-bin fast path
    (numpy/lib/_histograms_impl.py, `histogram`) explicitly and is additionally cross-checked
    against np.histogram in the tests.
  * Q/DQ arithmetic (ONNXRuntime QuantizeLinear/DequantizeLinear) is NOT pinned by any reference
    test or importable reference code: `fake_quant_qdq` follows the ONNX opset-13 operator spec
    ("parity unpinned" for that one function); `quant_acti` IS pinned (reference torch code).
"""
import math

import numpy as np

F32 = np.float32
OCTAV_CONST = 1 / (4 ** 8) / 3  # forward_net.py:327 -> 5.086263020833333e-06


# ------------------------------------------------------------------ a3: forward_get_minmax
def minmax(x):
    """forward_net.py:222-235 — per tensor per image `x.max()`, `x.min()` (np.float32; NaN propagates)."""
    x = np.asarray(x, F32).ravel()
    return F32(x.min()), F32(x.max())


def clip_minmax(mins, maxs):
    """basic_algorithm.py:21 — [np.min(minlist), np.max(maxlist)]."""
    return [F32(np.min(mins)), F32(np.max(maxs))]


# ------------------------------------------------------------------ a4: forward_get_hist
def hist_dmax(gmin, gmax):
    """forward_net.py:266-267 — data_max = max(np.max(maxlist), -np.min(minlist))  (python max: keeps
    the first argument unless the second is strictly greater)."""
    a, b = F32(gmax), F32(-F32(gmin))
    return b if b > a else a


def hist_edges(dmax, bins):
    """np.histogram(..., bins, (0, dmax)) outer edges and fp32 bin edges.

    _get_outer_edges: first == last -> (first - 0.5, last + 0.5).  _get_bin_edges: bin_type float32,
    np.linspace(first, last, bins + 1, dtype=float32) which evaluates in float32:
    step = fl32((last - first) / bins); edge[i] = fl32(fl32(i * step) + first); edge[bins] = last.
    """
    first, last = F32(0), F32(dmax)
    if not (math.isfinite(first) and math.isfinite(last)):
        raise ValueError(f"supplied range of [{first}, {last}] is not finite")
    if first == last:
        first, last = F32(first - F32(0.5)), F32(last + F32(0.5))
    delta = F32(last - first)
    step = F32(delta / F32(bins))
    i = np.arange(bins + 1, dtype=F32)
    if step == 0:  # linspace's denormal branch (gh-5437); histogram then raises "Too many bins"
        e = (i / F32(bins)) * delta + first
    else:
        e = i * step + first
    e = e.astype(F32)
    e[-1] = last
    if np.any(e[:-1] >= e[1:]):
        raise ValueError(f"Too many bins for data range. Cannot create {bins} finite-sized bins.")
    return first, last, e


def abs_hist(x, bins, dmax):
    """forward_net.py:268 — np.histogram(np.abs(x), int(bins), (0, data_max))[0], int64 counts.

    Restates numpy's uniform-bin fast path: keep first <= a <= last (drops NaN); index =
    trunc(fl32(fl32((a - first) / (last - first)) * bins)); == bins -> bins-1; a < edge[i] -> i-1;
    a >= edge[i+1] and i != bins-1 -> i+1.
    """
    bins = int(bins)
    a = np.abs(np.asarray(x, F32).ravel())
    first, last, e = hist_edges(dmax, bins)
    a = a[(a >= first) & (a <= last)]
    denom = F32(last - first)
    f = ((a - first).astype(F32) / denom).astype(F32) * F32(bins)
    idx = f.astype(np.int64)
    idx[idx == bins] -= 1
    idx[a < e[idx]] -= 1
    inc = (a >= e[idx + 1]) & (idx != bins - 1)
    idx[inc] += 1
    return np.bincount(idx, minlength=bins).astype(np.int64)


def hist_percentile(hist, gmin, gmax, bins, threshold):
    """basic_algorithm.py:40-53 — percentile clip from the summed |x| histogram.

    h = fl64(fl32(count)) / fl64(total); sequential fp64 accumulation from 0; first i with
    accum >= threshold -> clip = fl32(fl32(i + 0.5) * fl32(dmax / bins)); result
    [max(-clip, gmin), min(clip, gmax)]; never reached -> [gmin, gmax].  `dmax` here is
    max(-gmin, gmax) (line 42: argument order swapped w.r.t. forward_net.py:266, same value).
    """
    hist = np.asarray(hist)
    gmin, gmax = F32(gmin), F32(gmax)
    total = hist.sum()
    with np.errstate(all="ignore"):
        h = hist.astype(F32).astype(np.float64) / np.float64(total)
    a, b = F32(-gmin), gmax
    dmax = b if b > a else a
    accum = np.float64(0)
    for i in range(len(hist)):
        accum = accum + h[i]
        if accum >= threshold:
            clip = F32(F32(i + 0.5) * F32(dmax / F32(bins)))
            # python max(-clip, gmin): returns -clip unless gmin > -clip
            lo = gmin if gmin > F32(-clip) else F32(-clip)
            # python min(clip, gmax): returns clip unless gmax < clip
            hi = gmax if gmax < clip else clip
            return [F32(lo), F32(hi)]
    return [gmin, gmax]


# ------------------------------------------------------------------ a5: forward_net_octav
def octav_scale(x, unsigned=1):
    """forward_net.py:323-330 — OCTAV Newton-Raphson clip scale for one tensor of one image (np.float32).

    s0 = fl32(sum|x|) / count(|x| > 0); up to 20x: s' = fl32(sum_{|x|>s}|x|) / fl32(c/unsigned *
    count(|x|<=s) + count(|x|>s)) (the python-float denominator is cast to float32 before the divide,
    NEP 50); stop — KEEPING the previous s — when |s' - s| < 1e-6.
    """
    a = np.abs(np.asarray(x, F32).ravel())
    with np.errstate(all="ignore"):
        s = F32(a.sum() / F32(a[a > 0].size))
        for _ in range(20):
            gt = a > s
            denom = OCTAV_CONST / unsigned * int((a <= s).sum()) + int(gt.sum())
            s1 = F32(a[gt].sum() / F32(denom))
            if np.abs(F32(s1 - s)) < F32(1e-6):
                break
            s = s1
    return F32(s)


def octav_unsigned(data_min, dynamic_sym):
    """forward_net.py:319-322 — 4 when the platform's qi_params carries 'dynamic_sym' and |min| < 1e-6."""
    return 4 if (abs(float(data_min)) < 1e-6 and dynamic_sym) else 1


def octav_clip(s_list, min_list, max_list):
    """basic_algorithm.py:64-68 — [max(min_all, -mean(s)), min(max_all, mean(s))] with python max/min
    (a NaN mean therefore yields [min_all, max_all])."""
    with np.errstate(all="ignore"):
        m = np.array(s_list, F32).mean()
    dmax, dmin = np.array(max_list, F32).max(), np.array(min_list, F32).min()
    lo = F32(-m) if F32(-m) > dmin else dmin
    hi = m if m < dmax else dmax
    return [F32(lo), F32(hi)]


# ------------------------------------------------------------------ a9: weights
def rowwise_minmax(w, transpose=False):
    """basic_algorithm.py:84-90 — per output channel min/max over reshape(C, -1); ConvTranspose
    weights are transposed [1,0,2,3] first."""
    w = np.asarray(w)
    if transpose:
        w = w.transpose([1, 0, 2, 3])
    c = w.shape[0]
    w2 = w.reshape(c, -1)
    return w2.min(-1), w2.max(-1)


# ------------------------------------------------------------------ a11: get_qnode_by_param
def qparams(param, lo, hi):
    """quantize.py:111-194 — (scale fp32[], zero_point int8-wrapped[], q_min[], q_max[], symmetric).

    `lo`/`hi`: python/np scalars (per tensor) or 1-D arrays (per channel).  Reproduces: collapse to
    scalars when the platform is not per_channel (:120-122); dynamic_sym flip (:125-127); symmetric
    q = [-2^(b-1)+1, 2^(b-1)-1], scale = max|r| / q_max, 0 -> 1 (:128-143); asymmetric per-tensor
    (:146-162) and per-channel (:163-181); log_scale (:182-183); zero_point stored through np.int8
    (:185, wraps above 127).
    """
    b = param["bit_width"]
    per_channel = bool(param.get("per_channel", False))
    symmetric = param["symmetric"]
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    if not per_channel:
        lo, hi = np.float64(lo.min()), np.float64(hi.max())
        if param.get("dynamic_sym", False) and abs(lo - 0.0) < 1e-6:
            symmetric = False
    zero_point = np.array([0])
    if symmetric:
        ch = 1 if lo.ndim == 0 else lo.shape[0]
        q_min = [-2 ** (b - 1) + 1] * ch
        q_max = [2 ** (b - 1) - 1] * ch
        dmax = np.maximum(np.abs(lo), np.abs(hi))
        scale = np.atleast_1d(dmax / np.array(q_max if lo.ndim else q_max[0], np.float64))
        scale = np.where(scale == 0, 1.0, scale)
    elif lo.ndim == 0:
        dmin, dmax = min(0.0, float(lo)), max(0.0, float(hi))
        sc = (dmax - dmin) / (2 ** b - 1)
        if sc == 0.0:
            sc += 1.0
        zp = np.round(-dmin / sc)
        q_min, q_max = [int(-zp)], [int(2 ** b - 1 - zp)]
        scale, zero_point = np.array([sc]), np.array([zp])
    else:
        dmin, dmax = np.minimum(lo, 0.0), np.maximum(hi, 0.0)
        scale = (dmax - dmin) / (2 ** b - 1)
        scale = np.where(scale == 0, 1.0, scale)
        zp = (-dmin / scale).round()
        q_min = (-zp).astype(np.int32).tolist()
        q_max = (2 ** b - 1 - zp).astype(np.int32).tolist()
        zero_point = zp
    if param.get("log_scale", False):
        scale = 2 ** np.round(np.log2(scale))
    scale = np.asarray(scale, np.float64).astype(F32).ravel()
    with np.errstate(all="ignore"):
        zpw = np.broadcast_to(np.asarray(zero_point, np.float64), scale.shape).astype(np.int64)
        zpw = ((zpw + 128) % 256 - 128).astype(np.int8)  # np.full(..., dtype=np.int8) wrap
    return scale, zpw, list(q_min), list(q_max), bool(symmetric)


# ------------------------------------------------------------------ a12 / a14: fake quant
def quant_acti(x, scale, q_min, q_max):
    """weight_transform/ada_quant_layer.py:28-36 with prob = 1: round-half-even(x / scale), clamp to
    [q_min, q_max], times scale — all fp32."""
    x = np.asarray(x, F32)
    q = np.rint((x / F32(scale)).astype(F32))
    q = np.minimum(np.maximum(q, F32(q_min)), F32(q_max))
    return (q * F32(scale)).astype(F32)


def fake_quant_qdq(x, scale, zero_point, axis=None, signed=True):
    """quantize.py:197-239 builds QuantizeLinear -> DequantizeLinear; the arithmetic is the ONNX
    opset-13 spec executed by ONNXRuntime (third-party, absent here: PARITY UNPINNED):
    q = saturate(round_half_even(x / scale) + zp) to int8 [-128,127] / uint8 [0,255];
    y = (q - zp) * scale.  `axis` selects per-channel scale/zp."""
    x = np.asarray(x, F32)
    scale = np.asarray(scale, F32)
    zp = np.asarray(zero_point).astype(np.int64)
    if not signed:
        zp = zp & 0xFF  # int8-wrapped storage reinterpreted as UINT8 (quantize.py:185,205-206)
    if axis is not None and scale.size > 1:
        shp = [1] * x.ndim
        shp[axis] = -1
        scale = scale.reshape(shp)
        zp = zp.reshape(shp)
    lo, hi = (-128, 127) if signed else (0, 255)
    q = np.rint((x / scale).astype(F32)).astype(np.float64) + zp
    q = np.clip(q, lo, hi)
    return ((q - zp).astype(F32) * scale).astype(F32)


def cos_similarity(a, b):
    """utils.py:273-278 — sum(a*b) / sqrt(sum(a^2)) / sqrt(sum(b^2)); 0.0 when sum(a*b) == 0."""
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    d = np.sum(a * b)
    if d == 0:
        return 0.0
    return d / np.sqrt(np.square(a).sum()) / np.sqrt(np.square(b).sum())


def bias_correction_delta(fp_stack, q_stack, is_conv):
    """bias_correction.py:10-13 — stacks of per-image outputs [N, 1, C, H, W] (Conv) or [N, 1, C] (Gemm):
    np.squeeze(fp - q, axis=1).mean(axis=(0, 2, 3) or 0)."""
    d = np.stack(fp_stack, axis=0) - np.stack(q_stack, axis=0)
    return np.squeeze(d, axis=1).mean(axis=(0, 2, 3) if is_conv else 0)


def reduce_profiling_res(per_rank_layer, per_rank_model):
    """utils.py:386-412 — per_rank_layer: list of {tensor: cos} (or None with --model_type); per_rank_model: list of
    {output: [mean cos, min cos]}.  Ranks are weighted 1 / W in rank order; the minimum is taken over ranks."""
    w = float(len(per_rank_model))
    layer = {}
    if per_rank_layer is not None:
        layer = {k: v / w for k, v in per_rank_layer[0].items()}
        for d in per_rank_layer[1:]:
            for k, v in d.items():
                layer[k] += v / w
    model = {k: [v[0] / w, v[1]] for k, v in per_rank_model[0].items()}
    for d in per_rank_model[1:]:
        for k, v in d.items():
            model[k][0] += v[0] / w
            model[k][1] = min(model[k][1], v[1])
    return layer, model


# ------------------------------------------------------------------ a15: shard + merge
def shard_range(data_num, rank, world_size):
    """forward_net.py:207-209 — contiguous floor split; the remainder images are dropped."""
    rank_num = data_num // world_size
    return rank * rank_num, min((rank + 1) * rank_num, data_num)


def reduce_clip_val(per_rank, act_quant):
    """utils.py:326-345 — rank-0 merge of per-rank {name: [lo, hi]} (fp64 after the JSON round trip):
    minmax -> elementwise min / max; hist, mse -> sum_r(v_r / W) in rank order."""
    w = len(per_rank)
    out = {k: [np.float64(v[0]), np.float64(v[1])] for k, v in per_rank[0].items()}
    if act_quant != "minmax":
        for k in out:
            out[k][0] /= float(w)
            out[k][1] /= float(w)
    for r in range(1, w):
        for k, v in per_rank[r].items():
            if act_quant != "minmax":
                out[k][0] += v[0] / float(w)
                out[k][1] += v[1] / float(w)
            else:
                out[k] = [np.float64(min(v[0], out[k][0])), np.float64(max(v[1], out[k][1]))]
    return out
