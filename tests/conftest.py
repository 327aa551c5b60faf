import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# glibc writes its fatal diagnostics ("free(): invalid pointer", "malloc(): corrupted top size", stack smashing ...) to
# the controlling TERMINAL unless told otherwise -- on a GPU box they vanish and all a log shows is "Aborted".
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
# ... and a bare abort() in some library says nothing at all: libpct_hip.so then writes the native backtrace of the
# raising thread before Python's fault handler gets the signal (pct_api.hip: install_abort_trace).
os.environ.setdefault("PCT_ABORT_TRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # capturing is suspended while the session is configured: fd 2 is the real stderr here.  The library writes the
    # backtrace of an abort() to this copy -- what it writes to fd 2 during a test dies with pytest's capture file.
    try:
        os.environ["PCT_ABORT_TRACE"] = str(os.dup(2))
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))   # allow_pickle stays False
    return load


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the HIP extension and return the package modules."""
    import __graft_entry__ as ge
    ge.build()
    import pointCloudToolbox
    from point_cloud_toolbox_amd import _capi, shapes
    return dict(PointCloud=pointCloudToolbox.PointCloud, capi=_capi, shapes=shapes)


@pytest.fixture(scope="session")
def gpu(built):
    capi = built["capi"]
    if capi.device_count() < 1:
        pytest.fail("-m gpu tests need a GPU: pct_device_count() == 0 (no CPU fallback exists)")
    return built
