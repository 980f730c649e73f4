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


def has_oneread():
    """Was the library built with -DDPL_WITH_ONEREAD (the round-3 one-read OCTAV form: superseded, kept for A/B builds)?"""
    try:
        from dipoorlet_amd import _hip
        return bool(_hip.lib().dpl_octav_has_oneread())
    except Exception:   # noqa: BLE001
        return False


ONEREAD = pytest.param("oneread", marks=pytest.mark.skipif(not has_oneread(), reason="built without -DDPL_WITH_ONEREAD"))
