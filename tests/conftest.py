import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): oracle/liboracle.so through oracle/orc.py."""
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def native_libs():
    """Build (if stale and a compiler is present) and load the product libraries."""
    from simpleinfer_amd import _native
    if not (os.path.exists(_native.LIB_HIP_PATH) and os.path.exists(_native.LIB_HOST_PATH)):
        from simpleinfer_amd import build
        build.build_all()
    return _native.hip(), _native.host()


@pytest.fixture(scope="session")
def gpu(native_libs):
    """Fails (does not skip) when marked-gpu tests run without a device or without the HIP library."""
    from simpleinfer_amd import device_count
    n = device_count()
    assert n > 0, "gpu-marked test running without a HIP device"
    return n
