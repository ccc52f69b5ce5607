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
    """The ctypes binding of the product library.  If the in-tree .so has not been built yet (fresh checkout:
    `make` / `__graft_entry__.build()` not run), build it here -- hipcc cross-compiles gfx950 without a GPU."""
    from dxrexperiments_amd import capi
    if not os.environ.get("DXR_AMD_LIB") and not os.path.exists(os.path.join(ROOT, "dxrexperiments_amd", "lib", "libdxrexperiments_amd.so")):
        import subprocess
        subprocess.run(["make", "-C", ROOT, "-j4"], check=True, stdout=subprocess.DEVNULL)
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
