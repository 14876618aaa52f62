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
    # (before libgomp initialises) idle oracle threads sleep instead of spinning: the GPU box has far more cores than this container
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import orc as _orc
    # 16 threads, the reference's intra-op pool (engine_impl.cpp:133), not one per core: on the GPU box's 100+ cores the oracle's many small
    # parallel regions spent their time in fork / join (round 5: 118 CPU-minutes for an 8-minute suite; toy_yolo 20-46 s there against 1 s here)
    _orc.lib().orc_set_num_threads(min(16, os.cpu_count() or 1))
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


def pytest_terminal_summary(terminalreporter):
    """The element-wise mixed metric of every fp32 parity assertion of this run (tests/util.py assert_parity): the ten worst, so a run's log shows
    how far the suite sits from the 1e-4 bar under BOTH metrics; the full table goes to gpurun_out/mixed_metric.txt when that directory exists."""
    try:
        import util
    except Exception:  # noqa: BLE001
        return
    log = sorted(util.MIXED_LOG, key=lambda r: -r[2])
    if not log:
        return
    terminalreporter.write_line("fp32 parity under the element-wise metric max|d| / (|ref| + rms(ref)): %d assertions, worst %.3e (%s)" % (
        len(log), log[0][2], log[0][0]))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "mixed_metric.txt"), "w") as f:
            f.write("# what, max|diff|/max|ref|, max|d|/(|ref|+rms(ref))   (tests/util.py assert_parity; bar 1e-4 on both)\n")
            for what, e, m in log:
                f.write("%-70s %.3e %.3e\n" % (what, e, m))
