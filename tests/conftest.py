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
    build (csrc/hf_kernels.h HF_DBG_CHECK): at the end of the session the device-side violation records of this process must be empty on
    EVERY device it opened contexts on.  Child processes (bench.py ranks, host-I/O workers, cli workers) answer for their own records:
    capi.load() registers an exit hook under that library which ends the child with status 97 and a message on stderr, so the test that
    started it fails.  Limits: a check records and the access still executes (results of a violating run are not trustworthy); hosts that
    bind the C ABI without this Python layer call hf_debug_bounds_violations themselves.
    With the product library (no checks compiled in) this is a no-op."""
    yield
    from hopperrender_amd import capi
    if not capi.is_debug_bounds_build():
        return
    bad = {d: v for d, v in capi.debug_bounds_violations().items() if v[0]}
    assert not bad, f"out-of-range gather indices recorded, device: (count, [site, block, thread, line]) = {bad}"
