"""GPU: the device debug build (`python -m hopperrender_amd.build --debug-bounds`, -DHF_DEBUG_BOUNDS): every gather index of the kernels is
checked against its buffer and violations are recorded on the device (csrc/hf_kernels.h HF_DBG_CHECK).  The pool has no GPU
AddressSanitizer, so this is the device-side memory check (the reference's own out-of-range case is the single
reflection of calcDeltaSumsKernelSDR.h:86-95, which the HIP path clamps).

  * the self-test proves the checker fires: 64 out-of-range indices issued, 64 recorded, site 999;
  * a subset of the parity suite runs under the checking library in a child process (HF_LIB) -- tiny frames with huge offsets (the
    reference-UB case), ragged and strided sizes, the staged period warp with injected extreme flows at 2160p HDR, a batched
    chain -- and must finish with ZERO recorded violations and unchanged results (the tests' own assertions)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
from oracle import oracle

def violations(c, reset=0):
    n = C.c_uint32(0); first = (C.c_uint32 * 4)()
    capi.check(c._lib.hf_debug_bounds_violations(c._ctx, C.byref(n), first, reset), c._ctx)
    return n.value, list(first)

# 1. self-test: the checker fires
c = OpticalFlowCalcSDR(64, 96)
assert violations(c) == (0, [0, 0, 0, 0])
capi.check(c._lib.hf_debug_bounds_selftest(c._ctx), c._ctx)
n, first = violations(c, reset=1)
assert n == 64 and first[0] == 999, (n, first)
assert violations(c)[0] == 0
c.close()

# 2. the reference-UB corner: tiny frames, huge scalars (offsets larger than the frame), every radius
for hdr, H, W in ((0, 36, 64), (1, 48, 80), (0, 180, 320)):
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    sc = synth.Scene(H, W, bool(hdr), 5)
    fr = [sc.frame(k) for k in range(3)]
    g = oracle.make_geom(hdr, H, W)
    for R in (2, 5, 16):
        c = cls(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=R)
        for f in fr: c.updateFrame(f)
        c.calculateOpticalFlow()
        _, blur, tot, _ = oracle.calculate_optical_flow(fr[1], fr[2], g, R)
        assert np.array_equal(c.readBlurredFlow(1), blur) and c.m_totalFrameDelta == tot, (hdr, H, W, R)
        for mode in range(7):
            c.warpFrames(0.4, mode)
        c.copyFrame()
        assert violations(c)[0] == 0, (hdr, H, W, R, violations(c))
        c.close()

# 3. ragged + strided sizes
for hdr, H, W, si, so in ((0, 338, 600, 640, 608), (1, 338, 600, 608, 640), (0, 722, 1282, 0, 0)):
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    sc = synth.Scene(H, W & ~1, bool(hdr), 6, in_stride=si)
    c = cls(H, W & ~1, si, so, search_radius=8)
    for k in range(3): c.updateFrame(sc.frame(k))
    c.calculateOpticalFlow(); c.calculateOpticalFlow()
    for t in (0.0, 0.5, 1.0): c.warpFrames(t, 2)
    assert violations(c)[0] == 0, (hdr, H, W, violations(c))
    c.close()

# 4. the staged period warp (batch of 4 at 2160p HDR) with injected extreme flows: edge tiles, windows that do not fit, huge displacements
H, W, n = 2160, 3840, 4
g = oracle.make_geom(1, H, W)
sc = synth.Scene(H, W, True, 99)
fr = [sc.frame(k) for k in range(3)]
dev = []
for f in fr:
    b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
ms = [OpticalFlowCalcHDR(H, W, search_radius=5, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
rng = np.random.default_rng(3)
flows = [np.zeros((2, g.lh, g.lw), np.int16) for _ in range(4)]
flows[0][0], flows[0][1] = 9, -5
flows[1][:] = rng.integers(-40, 41, size=flows[1].shape)
flows[2][0], flows[2][1] = 30000, -30000                  # far beyond the frame: every coordinate is clamped
flows[3][0] = np.linspace(-600, 600, g.lw).astype(np.int16)[None, :]; flows[3][1] = np.linspace(-300, 300, g.lh).astype(np.int16)[:, None]
for i, m in enumerate(ms):
    for k in range(3): m.updateFrameDeviceRef(dev[k].ptr)
    m.sync(); m.writeBlurredFlow(0, flows[i])
batch = FlowBatch(ms)
ts = [0.0, 0.1988, 0.5, 0.7992, 0.998]
outs = [[DeviceBuffer(ms[0].output_frame_bytes) for _ in ts] for _ in range(n)]
for mode in (2, 0, 1):
    batch.interpolatePeriod([ts] * n, [[b.ptr for b in o] for o in outs], mode)
    batch.sync()
for i in (0, 2):   # and the results are the oracle's
    want = oracle.warp_frames(fr[0], fr[1], flows[i], g, np.float32(ts[3]), 2)
    batch.interpolatePeriod([ts] * n, [[b.ptr for b in o] for o in outs], 2); batch.sync()
    assert np.array_equal(outs[i][3].download(np.uint16), want), i
# 5. a batched chain + deferred planes
for k in range(3):
    batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, [ts] * n, [[b.ptr for b in o] for o in outs], 2))
batch.sync()
assert violations(ms[0])[0] == 0, violations(ms[0])
batch.close()
print("DEBUG-BOUNDS-OK")
"""


def test_parity_subset_under_the_bounds_checking_build(native_lib):
    from hopperrender_amd import build
    dbg = build.build_flow(debug_bounds=True)
    env = dict(os.environ, HF_LIB=dbg)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0 and "DEBUG-BOUNDS-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_product_build_has_no_checks(native_lib):
    import ctypes as C
    from hopperrender_amd import capi
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    if os.environ.get("HF_LIB"):
        pytest.skip("HF_LIB selects another build")
    c = OpticalFlowCalcSDR(64, 96)
    n = C.c_uint32(0)
    assert c._lib.hf_debug_bounds_violations(c._ctx, C.byref(n), None, 0) == capi.HF_ERR_STATE
    assert c._lib.hf_debug_bounds_selftest(c._ctx) == capi.HF_ERR_STATE
    c.close()
