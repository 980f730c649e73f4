"""torch.ops.dipoorlet.* (dipoorlet_amd/torch_ops.py): registered, CPU calls refused (no fallback),
GPU results equal to the oracle."""
import numpy as np
import pytest
import torch

import dipoorlet_amd.torch_ops  # noqa: F401
from oracle import np_oracle as O

NAMES = ("minmax", "minmax_batched", "abs_hist_", "hist_percentile", "octav", "rowwise_minmax", "fake_quant")


def test_registered_and_no_cpu_path():
    for n in NAMES:
        assert hasattr(torch.ops.dipoorlet, n), n
    with pytest.raises(NotImplementedError):
        torch.ops.dipoorlet.minmax(torch.zeros(8))
    with pytest.raises(NotImplementedError):
        torch.ops.dipoorlet.fake_quant(torch.zeros(8), torch.ones(1), torch.zeros(1, dtype=torch.int32), 0, -128, 127)


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
    q = torch.ops.dipoorlet.fake_quant(xd, torch.tensor([0.05], device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"),
                                       0, -128, 127).cpu().numpy()
    assert np.array_equal(q, O.fake_quant_qdq(x, np.float32(0.05), 0))
