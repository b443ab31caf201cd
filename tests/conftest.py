import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
