import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "two_forwards: compares the results of two runs of the library forward (two sessions, two "
                                       "schedules, two processes): runs under the library's deterministic algorithms, so that the "
                                       "comparison is exact instead of a bound on library noise")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Collection order under `-x`: a failure stops the run, so what holds the hand-written kernels to the oracle and to the reference's
# fixtures runs FIRST, and the end-to-end files — whole CLI runs over library convolutions, child processes, rendezvous — LAST
# (alphabetical order put test_baseline_configs / test_cli_e2e in front of test_hip_parity: one e2e failure hid every kernel test).
_ORDER = ["test_capi_load", "test_hip_parity", "test_round_parity", "test_torch_ops", "test_aux_golden", "test_sparse_quant",
          "test_baseline_configs", "test_calibration_e2e", "test_multirank_gpu",
          "test_weight_transforms_e2e", "test_round_e2e", "test_cli_e2e"]


def _rank(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    tier = _ORDER.index(mod) if mod in _ORDER else len(_ORDER) // 2
    # inside a kernel-parity file the tests that run a CLI (test_sparse_quant's end-to-end run) go behind its kernel tests
    late = 1 if ("cli" in item.name or "e2e" in item.name) and tier < _ORDER.index("test_baseline_configs") else 0
    return (tier, late)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)       # (stable: the order inside a file is kept)


@pytest.fixture(autouse=True)
def _deterministic_library_where_two_forwards_are_compared(request, monkeypatch):
    """MIOpen's default choice for a 3 x 3 stride-2 convolution on gfx950 (`igemm_fwd_gtcx35_nhwc_fp32_*`, solver
    ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC) splits the reduction over workgroups and adds the partial sums into a zeroed
    output with fp32 atomics: two calls on the same input differ by 1e-6 (scripts/conv_repro_probe.py), a
    fake-quantised forward turns that into a flipped rounding step now and then, a flipped step moves a corrected bias by 5e-4.  A
    test marked `two_forwards` compares two runs of the forward — it runs under the library's deterministic algorithms
    (torch.backends.cudnn.deterministic; the CLI's DPL_DETERMINISTIC=1, which child processes inherit): every convolution is then
    bit-reproducible from call to call, from session to session and from process to process, and the comparison is exact."""
    if request.node.get_closest_marker("two_forwards") is None:
        yield
        return
    import torch
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    monkeypatch.setenv("DPL_DETERMINISTIC", "1")
    try:
        yield
    finally:
        torch.backends.cudnn.deterministic = old
