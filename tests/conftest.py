import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _deterministic_convolutions_where_two_schedules_are_compared(request, monkeypatch):
    """MIOpen's default convolution kernels are not bit-reproducible from call to call (1e-6 relative between two forwards of one
    session).  A test that runs `--bc` twice — two schedules, one answer — would now and then see a rounding step of the
    fake-quantised forward flip and a bias move by 5e-4: such tests run under the library's deterministic algorithms (the CLI's
    DPL_DETERMINISTIC=1; child processes inherit it)."""
    if "bias_correction" not in request.node.name:
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
