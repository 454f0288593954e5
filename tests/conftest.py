import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """The product library; built in-tree if stale.  No fallback: failure to load fails the test."""
    from hopperrender_amd import build, capi
    build.build_all()
    return capi.load()


@pytest.fixture(scope="session", autouse=True)
def _no_bounds_violations_under_the_debug_library():
    """`HF_LIB=hopperrender_amd/lib/libhopperflow_dbg.so python -m pytest tests -m gpu` runs the whole GPU suite on the bounds-checking
    build (csrc/hf_kernels.h HF_DBG_CHECK): at the end of the session the device-side violation records of this process must be empty.
    With the product library (no checks compiled in) this is a no-op."""
    yield
    if "libhopperflow_dbg" not in os.environ.get("HF_LIB", ""):
        return
    import ctypes as C
    from hopperrender_amd import capi
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    c = OpticalFlowCalcSDR(64, 96)
    n = C.c_uint32(0)
    first = (C.c_uint32 * 4)()
    capi.check(c._lib.hf_debug_bounds_violations(c._ctx, C.byref(n), first, 0), c._ctx)
    c.close()
    assert n.value == 0, f"{n.value} out-of-range gather indices recorded; first: site {first[0]}, block {first[1]}, thread {first[2]}, line {first[3]}"
