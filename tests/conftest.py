import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "content-aware-gan-compression_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["launch_policy", "f4_forced"])
def wino4_policy(request):
    """Which Winograd kernel an F(4x4)-eligible layer's launch takes: the library's per-launch choice (F(4x4,3x3) when its grid of
    64-channel workgroups fills the chip, else the F(2x2,3x3) packing the layer also carries — what the B = 1-2 per-layer tests
    get), or F(4x4) forced (`cagc_set_tuning("wino4_min_wgs", 0)`: the kernel the same layer runs on at the bench's batch 16)."""
    from cagc import _lib
    if request.param == "f4_forced":
        with _lib.tuning(wino4_min_wgs=0):      # restores the PREVIOUS value (a CAGC_WINO4_MIN_WGS override survives)
            yield request.param
    else:
        yield request.param


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
