"""torch.ops.dipoorlet.* (dipoorlet_amd/torch_ops.py): registered, CPU calls refused (no fallback), GPU results
equal to the oracle — per tensor and over a whole batch's tensor set in one launch —, and from the SECOND call with a given
tensor-set geometry on: no device allocation besides the outputs, no host synchronisation."""
import warnings

import numpy as np
import pytest
import torch

import dipoorlet_amd.torch_ops  # noqa: F401
from oracle import np_oracle as O

NAMES = ("minmax", "minmax_batched", "abs_hist_", "abs_hist_batched_", "hist_percentile", "octav", "octav_batched", "rowwise_minmax",
         "fake_quant", "fake_quant_relu", "fake_quant_add_relu", "fake_quant_set")


def test_registered_and_no_cpu_path():
    for n in NAMES:
        assert hasattr(torch.ops.dipoorlet, n), n
    with pytest.raises(NotImplementedError):
        torch.ops.dipoorlet.minmax(torch.zeros(8))
    with pytest.raises(NotImplementedError):
        torch.ops.dipoorlet.fake_quant(torch.zeros(8), torch.ones(1), torch.zeros(1, dtype=torch.int32), 0, -128, 127)
    with pytest.raises(NotImplementedError):
        torch.ops.dipoorlet.octav_batched([torch.zeros(2, 8)], False)


@pytest.mark.gpu
def test_ops_against_oracle():
    from _cases import make_tensor
    x = make_tensor("relu", 25088, 4)
    xd = torch.from_numpy(x).cuda()
    mm = torch.ops.dipoorlet.minmax(xd).cpu().numpy()
    lo, hi = O.minmax(x)
    assert mm[0] == lo and mm[1] == hi
    y = make_tensor("normal", 2048, 5)
    mins = torch.full((2,), float("inf"), device="cuda")
    maxs = torch.full((2,), float("-inf"), device="cuda")
    torch.ops.dipoorlet.minmax_batched([xd, torch.from_numpy(y).cuda()], mins, maxs)
    torch.ops.dipoorlet.minmax_batched([xd * 2, torch.from_numpy(y).cuda()], mins, maxs)
    assert mins.cpu().tolist() == [float(min(lo, 2 * lo)), float(y.min())]
    assert maxs.cpu().tolist() == [float(2 * hi), float(y.max())]
    dmax = O.hist_dmax(lo, hi)
    hist = torch.zeros(2048, dtype=torch.int64, device="cuda")
    torch.ops.dipoorlet.abs_hist_(xd, float(dmax), 2048, hist)
    torch.ops.dipoorlet.abs_hist_(xd, float(dmax), 2048, hist)
    h = O.abs_hist(x, 2048, dmax)
    assert np.array_equal(hist.cpu().numpy(), 2 * h)
    clip = torch.ops.dipoorlet.hist_percentile(hist, float(lo), float(hi), 0.99999).cpu().numpy()
    ref = O.hist_percentile(2 * h, lo, hi, 2048, 0.99999)
    assert clip[0] == ref[0] and clip[1] == ref[1]
    s = torch.ops.dipoorlet.octav(xd, False).cpu().numpy()
    es = O.octav_scale(x, 1)
    assert s[0] == pytest.approx(float(es), rel=1e-5) and s[1] == lo and s[2] == hi
    w = make_tensor("normal", 64 * 27, 6).reshape(64, 27)
    rlo, rhi = torch.ops.dipoorlet.rowwise_minmax(torch.from_numpy(w).cuda())
    assert np.array_equal(rlo.cpu().numpy(), w.min(1)) and np.array_equal(rhi.cpu().numpy(), w.max(1))
    sc, zp = torch.tensor([0.05], device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    q = torch.ops.dipoorlet.fake_quant(xd, sc, zp, 0, -128, 127).cpu().numpy()
    assert np.array_equal(q, O.fake_quant_qdq(x, np.float32(0.05), 0))
    z = make_tensor("normal", 25088, 9)
    zd = torch.from_numpy(z).cuda()
    q = torch.ops.dipoorlet.fake_quant_relu(zd, sc, zp, 0, -128, 127).cpu().numpy()
    assert np.array_equal(q, O.fake_quant_qdq(np.maximum(z, np.float32(0)), np.float32(0.05), 0))
    q = torch.ops.dipoorlet.fake_quant_add_relu(zd, xd, sc, zp, 0, -128, 127).cpu().numpy()
    assert np.array_equal(q, O.fake_quant_qdq(np.maximum((z + x).astype(np.float32), np.float32(0)), np.float32(0.05), 0))


# One batch of B images of a small "network": the golden distributions (tests/_cases.py) incl. the adversarial ones — exact zeros,
# one huge outlier, values on the bin edges, a dead tensor, a tensor whose minimum is within 1e-6 of zero (dynamic_sym)
SET = (("normal", 25088), ("relu", 150528), ("laplace", 1000), ("uniform", 2048), ("zeros", 512), ("spike", 25088), ("edges", 25088),
       ("neg_only", 1000), ("tiny", 2048))
B = 3


def _batch(seed0):
    from _cases import make_tensor
    host = [np.stack([make_tensor(kind, n, seed0 + 100 * t + b) for b in range(B)]) for t, (kind, n) in enumerate(SET)]
    return host, [torch.from_numpy(h).cuda() for h in host]


@pytest.mark.gpu
def test_batched_ops_against_oracle():
    """minmax_batched / abs_hist_batched_ / octav_batched / fake_quant_set over every tensor of a batch in one launch each, two
    batches accumulated: the reference's loops over `ort_outputs` (forward_net.py:220-235, 265-280, 314-340) through the oracle."""
    T = len(SET)
    sets = [_batch(1), _batch(7)]
    mins = torch.full((T,), float("inf"), device="cuda")
    maxs = torch.full((T,), float("-inf"), device="cuda")
    for _, dev in sets:
        torch.ops.dipoorlet.minmax_batched(dev, mins, maxs)
    lo = [min(O.minmax(h[t])[0] for h, _ in sets) for t in range(T)]
    hi = [max(O.minmax(h[t])[1] for h, _ in sets) for t in range(T)]
    assert mins.cpu().tolist() == [float(v) for v in lo] and maxs.cpu().tolist() == [float(v) for v in hi]
    for bins in (2048, 1000):
        hist = torch.zeros(T, bins, dtype=torch.int64, device="cuda")
        for _, dev in sets:
            torch.ops.dipoorlet.abs_hist_batched_(dev, mins, maxs, bins, hist)
        got = hist.cpu().numpy()
        for t in range(T):
            want = sum(O.abs_hist(h[t].reshape(-1), bins, O.hist_dmax(lo[t], hi[t])) for h, _ in sets)
            assert np.array_equal(got[t], want), (bins, SET[t])
            clip = torch.ops.dipoorlet.hist_percentile(hist[t], float(lo[t]), float(hi[t]), 0.99999).cpu().numpy()
            ref = np.asarray(O.hist_percentile(want, lo[t], hi[t], bins, 0.99999), np.float32)
            assert clip.view(np.uint32).tolist() == ref.view(np.uint32).tolist(), (bins, SET[t])
    for dyn in (False, True):
        for h, dev in sets:
            rows = torch.ops.dipoorlet.octav_batched(dev, dyn).cpu().numpy()
            assert rows.shape == (B, T, 3)
            for t in range(T):
                for b in range(B):
                    mn, mx = O.minmax(h[t][b])
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        s = O.octav_scale(h[t][b], O.octav_unsigned(mn, dyn))
                    assert rows[b, t, 1] == mn and rows[b, t, 2] == mx
                    assert np.allclose(rows[b, t, 0], s, rtol=1e-5, atol=1e-5, equal_nan=True), (dyn, SET[t], b, rows[b, t, 0], s)
    # fake_quant_set: per tensor on the int8 grid, per channel (axis 1 of [B, C, inner]) on the uint8 grid with zero points
    h, dev = sets[0]
    rng = np.random.default_rng(3)
    views, scales, zps, inner, qlo, qhi, want = [], [], [], [], [], [], []
    for t, (kind, n) in enumerate(SET):
        C_ = 8 if (t % 2 and n % 8 == 0) else 1
        sc = rng.uniform(0.01, 0.1, C_).astype(np.float32)
        zp = rng.integers(0, 255, C_).astype(np.int32) if C_ > 1 else np.zeros(1, np.int32)
        views.append(dev[t].view(B, C_, n // C_))
        scales.append(torch.from_numpy(sc).cuda())
        zps.append(torch.from_numpy(zp).cuda())
        inner.append(n // C_)
        qlo.append(0 if C_ > 1 else -128)
        qhi.append(255 if C_ > 1 else 127)
        want.append(O.fake_quant_qdq(h[t].reshape(B, C_, n // C_), sc, zp, axis=1 if C_ > 1 else None, signed=C_ == 1))
    for _ in range(2):
        ys = torch.ops.dipoorlet.fake_quant_set(views, scales, zps, inner, qlo, qhi)
        for t in range(T):
            assert np.array_equal(ys[t].cpu().numpy().view(np.uint32), want[t].view(np.uint32)), SET[t]


@pytest.mark.gpu
def test_second_call_allocates_only_outputs_and_does_not_synchronise():
    """"no hidden syncs, no allocation inside except outputs".  Every op once (plans, accumulators, OCTAV workspace
    are built), then again on NEW input tensors of the same geometry under torch.cuda.set_sync_debug_mode('error') with the
    caching allocator's allocation count read before and after: the difference is the op's outputs, nothing else."""
    T = len(SET)
    _, first = _batch(11)
    _, second = _batch(12)
    _, third = _batch(13)
    mins = torch.full((T,), float("inf"), device="cuda")
    maxs = torch.full((T,), float("-inf"), device="cuda")
    hist = torch.zeros(T, 2048, dtype=torch.int64, device="cuda")
    h1 = torch.zeros(2048, dtype=torch.int64, device="cuda")
    sc, zp = torch.tensor([0.05], device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    scales, zps = [sc] * T, [zp] * T
    inner, qlo, qhi = [1] * T, [-128] * T, [127] * T
    calls = {
        "minmax": (lambda xs: torch.ops.dipoorlet.minmax(xs[0]), 1),
        "minmax_batched": (lambda xs: torch.ops.dipoorlet.minmax_batched(xs, mins, maxs), 0),
        "abs_hist_": (lambda xs: torch.ops.dipoorlet.abs_hist_(xs[0], 5.0, 2048, h1), 0),
        "abs_hist_batched_": (lambda xs: torch.ops.dipoorlet.abs_hist_batched_(xs, mins, maxs, 2048, hist), 0),
        "hist_percentile": (lambda xs: torch.ops.dipoorlet.hist_percentile(h1, -1.0, 5.0, 0.99999), 1),
        "octav": (lambda xs: torch.ops.dipoorlet.octav(xs[1], False), 1),
        "octav_batched": (lambda xs: torch.ops.dipoorlet.octav_batched(xs, False), 1),
        "fake_quant": (lambda xs: torch.ops.dipoorlet.fake_quant(xs[0], sc, zp, 0, -128, 127), 1),
        "fake_quant_relu": (lambda xs: torch.ops.dipoorlet.fake_quant_relu(xs[0], sc, zp, 0, -128, 127), 1),
        "fake_quant_add_relu": (lambda xs: torch.ops.dipoorlet.fake_quant_add_relu(xs[0], xs[0], sc, zp, 0, -128, 127), 1),
        "fake_quant_set": (lambda xs: torch.ops.dipoorlet.fake_quant_set(xs, scales, zps, inner, qlo, qhi), T),
    }
    for name, (fn, _) in calls.items():          # first calls: everything cached is built here
        fn(first)
    torch.cuda.synchronize()
    keep = []
    for xs in (second, third):                   # later calls, on tensors at other addresses
        for name, (fn, n_out) in calls.items():
            before = torch.cuda.memory_stats()["allocation.all.allocated"]
            torch.cuda.set_sync_debug_mode("error")
            try:
                keep.append(fn(xs))
            finally:
                torch.cuda.set_sync_debug_mode("default")
            after = torch.cuda.memory_stats()["allocation.all.allocated"]
            assert after - before == n_out, (name, after - before, n_out)
    torch.cuda.synchronize()
