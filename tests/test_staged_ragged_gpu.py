"""GPU: the workgroup-staged period warp (warp_wg_kernel) on frames whose rows do not end on a 16-byte thread, and whose chroma plane has
an odd number of rows.  Phase C of the staged body stores 16 bytes per lane unconditionally, so a wave whose last lane hangs
over the row's end has to take the generic body: the reference writes x < W only (warpFrameKernelHDR.h:116-120) and an integrator that
renders into a sub-rectangle of a larger surface must find the columns right of W untouched.  Every output is judged by the pinned oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T5 = [0.1988, 0.3996, 0.5984, 0.7992, 0.998]


@pytest.mark.parametrize("hdr,H,W,SO,n", [
    (1, 2160, 3836, 3904, 16),     # W % 8 == 4: lane 15 of the last tile column owns elements 3832 .. 3839, four of them beyond the row
    (1, 2162, 3840, 3904, 4),      # H / 2 = 1081 chroma rows: the last row group of the chroma plane has one row
    (1, 2162, 3836, 3840, 4),      # both
    (0, 2160, 3832, 3904, 8),      # 8-bit frames at 2160p (16 elements per thread): W % 16 == 8
])
def test_staged_warp_leaves_the_stride_padding_alone(native_lib, hdr, H, W, SO, n):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    cls, dt = (OpticalFlowCalcHDR, np.uint16) if hdr else (OpticalFlowCalcSDR, np.uint8)
    SI, R = W + (8 if hdr else 16), 16
    g = oracle.make_geom(hdr, H, W, SI, SO)
    sc = synth.Scene(H, W, bool(hdr), 77, in_stride=SI)
    frames = [sc.frame(k) for k in range(3)]
    dev = []
    for f in frames:
        b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    members = [cls(H, W, SI, SO, 8, 6, 0.0, 255.0, 270, search_radius=R, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for _ in range(n)]
    batch = FlowBatch(members)
    F_out = members[0].output_frame_bytes
    sentinel = 0xA5A5 if hdr else 0xA5
    outs = [[DeviceBuffer(F_out) for _ in range(5)] for _ in range(n)]
    for o in outs:
        for b in o:
            b.upload(np.full(F_out // dt().itemsize, sentinel, dt))
    plans = [T5[i % 5:] + T5[:i % 5] for i in range(n)]
    _, flow, _, oob = oracle.calculate_optical_flow(frames[0], frames[1], g, R)
    assert oob == 0
    try:
        oracle.set_flavour(1, 1, None)
        for k in range(2):
            batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, None, None, calculate_flow=(k == 1)))
        # period 2: the outputs between frames 0 and 1 with the flow (f0, f1) -- one fused launch for the whole batch
        batch.runPeriod(batch.preparePeriod([dev[2].ptr] * n, plans, [[b.ptr for b in o] for o in outs], 2))
        batch.sync()
        want = {}
        for i, m in enumerate(members):
            assert np.array_equal(m.readBlurredFlow(0), flow), i
            for j, t in enumerate(plans[i]):
                if t not in want:
                    want[t] = oracle.warp_frames(frames[0], frames[1], flow, g, np.float32(t), 2).reshape(-1, SO)
                got = outs[i][j].download(dt).reshape(-1, SO)
                assert np.array_equal(got[:, :W], want[t][:, :W]), (i, j, t)
                assert (got[:, W:] == sentinel).all(), ("padding columns written", i, j, int((got[:, W:] != sentinel).sum()))
    finally:
        batch.close()
        for m in members:
            m.close()
        for b in dev + [x for o in outs for x in o]:
            b.free()
