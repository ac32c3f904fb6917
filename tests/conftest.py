import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("filterwarnings", "ignore:.*trapz.*:DeprecationWarning")


def _gpu_count():
    try:
        from baryonification_amd import _lib
        return _lib.load().bfgx_device_count()
    except Exception:
        return 0


@pytest.fixture(scope='session')
def gpu():
    """GPU tests must FAIL (not skip) when selected with -m gpu on a box without the HIP path."""
    from baryonification_amd import _lib
    n = _lib.load().bfgx_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need the MI355X"
    return n
