import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the reference tree (build container only)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _gpu_expected():
    """A GPU box: the kernel driver's device node is there (or RT_EXPECT_GPU=1 says so).  On such a machine a GPU test
    must not be skipped because the runtime fails to see the card -- that would read as green."""
    return os.environ.get("RT_EXPECT_GPU") == "1" or os.path.exists("/dev/kfd")


@pytest.fixture(autouse=True)
def _gpu_must_be_visible(request):
    if request.node.get_closest_marker("gpu") and not _gpu_available() and _gpu_expected():
        pytest.fail("a GPU is expected on this machine (/dev/kfd or RT_EXPECT_GPU=1) but torch.cuda.is_available() is False")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/rt_oracle.c), compiled on demand with gcc."""
    from oracle import rt_oracle
    rt_oracle.build()
    rt_oracle.lib()
    return rt_oracle


@pytest.fixture(scope="session")
def renderer():
    """One GPU context for the whole session; the product path has no CPU fallback."""
    if not _gpu_available():
        if _gpu_expected():
            pytest.fail("a GPU is expected on this machine (/dev/kfd or RT_EXPECT_GPU=1) but torch.cuda.is_available() is False")
        pytest.skip("no GPU visible")
    from raytracer_amd import runtime
    r = runtime.Renderer(0)
    yield r
    r.close()
