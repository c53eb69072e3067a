import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_before_anything_loads():
    """Bring both shared libraries up to date BEFORE any test dlopens one of them: a rebuild of
    libmms_hip.so after libmms_caffe.so has mapped the old file would leave two copies of the
    library (and of its process-wide settings) in the process."""
    from mms_answer_selection_amd import build
    build.build_all()


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def hiplib():
    """Builds (if stale) and loads the C-ABI library; never skips on failure."""
    from mms_answer_selection_amd import build, capi
    build.build_all()
    return capi.lib()


@pytest.fixture(autouse=True)
def _reference_backward_rounding(request):
    """GPU tests assert the reference's BITS for the Euclidean backward term unless they say
    otherwise: select MMS_EUCLID_BWD_REFERENCE around each of them (include/mms.h).  The product
    default (fp32 arithmetic, <= 2 ulp) is exercised by tests/test_gpu_bwd_modes.py, smoke() and
    bench.py."""
    if "gpu" not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from mms_answer_selection_amd import capi
    capi.set_euclid_backward_mode("reference")
    yield
    capi.set_euclid_backward_mode("fp32")
