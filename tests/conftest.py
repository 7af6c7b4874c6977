import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        from multiview_stitcher_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip_device():
    """GPU tests call through the C ABI; no device -> the test fails loudly (never skips to a CPU path)."""
    from multiview_stitcher_amd import _lib

    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    _lib.init(0)
    return 0
