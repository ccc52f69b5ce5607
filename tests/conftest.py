import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    from dxrexperiments_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def gpu(capi):
    """A context on GPU 0.  GPU tests must fail loudly, not skip, when the HIP path is unusable."""
    n = capi.device_count()
    assert n > 0, "no HIP device visible"
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
