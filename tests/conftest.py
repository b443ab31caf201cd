import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Load order matters on ROCm: PyTorch bundles its own copy of the HIP runtime, librtw_hip.so links the system one
# (/opt/rocm).  When torch is imported first the dynamic linker hands librtw_hip.so the runtime that is already loaded (same
# soname) and both share one; the other way round torch brings up a SECOND runtime in the process, whose device enumeration
# fails ("No HIP GPUs are available", gpurun_out/r02_debug_torch.log).  Tests that hand torch tensors to the library
# (device output pointers, bench rehearsals) therefore need torch loaded before the first rtw call.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing for a few tests only
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rtw():
    import rtw_amd
    return rtw_amd


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def gpu(rtw):
    """A Renderer on cuda:0; fails (does not skip) when the HIP path is unavailable on a GPU run."""
    assert rtw.device_count() > 0, "no HIP device visible: -m gpu tests need the MI355X"
    r = rtw.Renderer(0)
    yield r
    r.close()
