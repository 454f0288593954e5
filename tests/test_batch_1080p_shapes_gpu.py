"""GPU: the timed launch of BASELINE configs 2 and 4 -- hf_batch_run_period on batches of 12 and 16 pairs of 1920 x 1080 SDR frames (the
period warp of 8-bit frames up to 1080p) -- in the shapes the other batched 1080p tests leave out (VERDICT r4 "what's weak" 1): padded
strides (in_stride 2048 / out_stride 1984: opticalFlowCalcSDR.cpp:212-213, :160-161 pass both to every kernel) and non-default output
levels (16/235 next to 0/255), every output against the pinned oracle with the padding columns untouched; and the same call at default
strides directly against the reference's golden frames (sdr_1080p SHA-256, incl. warp_m2_t0.5_lv16_235)."""
import numpy as np
import pytest

from helpers import Golden, sha

pytestmark = pytest.mark.gpu

PLANS = [[0.3996, 0.7992], [0.1988, 0.5984, 0.998], [0.0, 0.3996, 0.7992]]     # 24 -> 60 fps: two or three outputs per source period
LEVELS = [(0.0, 255.0), (16.0, 235.0)]


@pytest.mark.parametrize("n", [12, 16])
def test_strided_1080p_batches_with_levels(native_lib, n):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcSDR
    from oracle import oracle
    H, W, SI, SO, R = 1080, 1920, 2048, 1984, 16
    g = oracle.make_geom(0, H, W, SI, SO)
    sc = synth.Scene(H, W, False, 4242, in_stride=SI)
    frames = [sc.frame(k) for k in range(4)]
    dev = []
    for f in frames:
        b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    members = [OpticalFlowCalcSDR(H, W, SI, SO, 8, 6, *LEVELS[i % 2], 270, search_radius=R, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for i in range(n)]
    batch = FlowBatch(members)
    F_out = members[0].output_frame_bytes
    assert F_out == (H + H // 2) * SO
    outs = [[DeviceBuffer(F_out) for _ in range(3)] for _ in range(n)]
    for o in outs:
        for b in o:
            b.upload(np.full(F_out, 0xA5, np.uint8))
    optr = [[b.ptr for b in o] for o in outs]
    plans = [PLANS[i % 3] for i in range(n)]
    flows = {}
    for k in (1, 2):
        _, flows[k], _, oob = oracle.calculate_optical_flow(frames[k - 1], frames[k], g, R)
        assert oob == 0
    try:
        oracle.set_flavour(1, 1, None)       # 0/255 and 16/235 are pinned by tests/golden/levels_ramp.npz with the oracle's default reciprocal
        batch.runPeriod(batch.preparePeriod([d.ptr for d in [dev[0]] * n], None, None, calculate_flow=False))
        batch.runPeriod(batch.preparePeriod([dev[1].ptr] * n, None, None))
        for k in (2, 3):
            batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, plans, optr, 2))
            batch.sync()
            want = {}
            for i, m in enumerate(members):
                assert np.array_equal(m.readBlurredFlow(0), flows[k - 1]), (k, i)
                for j, t in enumerate(plans[i]):
                    key = (t, i % 2)
                    if key not in want:
                        want[key] = oracle.warp_frames(frames[k - 2], frames[k - 1], flows[k - 1], g, np.float32(t), 2, *LEVELS[i % 2]).reshape(-1, SO)
                    got = outs[i][j].download(np.uint8).reshape(-1, SO)
                    assert np.array_equal(got[:, :W], want[key][:, :W]), (k, i, j, t)
                    assert (got[:, W:] == 0xA5).all(), (k, i, j)
    finally:
        batch.close()
        for m in members:
            m.close()
        for b in dev + [x for o in outs for x in o]:
            b.free()


@pytest.mark.parametrize("n", [12, 16])
def test_1080p_batches_match_the_reference_frames(native_lib, n):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcSDR
    g = Golden("sdr_1080p")
    frames = g.frames()
    key = "R16_d8_n6"
    R, delta, nb = g.params(key)
    dev = []
    for f in frames:
        b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    lv = lambda i: (16.0, 235.0) if i % 4 == 3 else (0.0, 255.0)
    members = [OpticalFlowCalcSDR(g.case["H"], g.case["W"], g.case["si"], g.case["so"], delta, nb, *lv(i), g.case.get("max_res", 270),
                                  search_radius=R, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for i in range(n)]
    batch = FlowBatch(members)
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in range(3)] for _ in range(n)]
    optr = [[b.ptr for b in o] for o in outs]
    plans = [[0.5, 0.25, 0.75] if i % 4 == 3 else PLANS[i % 3] for i in range(n)]
    for k in range(4):
        if k < 2:
            batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, None, None, calculate_flow=False))
        else:
            batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, plans, optr, 2))
    batch.sync()
    names = g.frame_names(key)
    n_checked = 0
    for i, m in enumerate(members):
        assert np.array_equal(m.readBlurredFlow(0), g.arr(key, "blur_a")), i
        assert np.array_equal(m.readBlurredFlow(1), g.arr(key, "blur_b")), i
        for j, t in enumerate(plans[i]):
            fname = f"warp_m2_t{t}" + ("_lv16_235" if i % 4 == 3 else "")
            if fname in names:
                assert sha(outs[i][j].download(np.uint8)) == g.frame_sha(key, fname), (i, fname)
                n_checked += 1
    assert n_checked >= n
    batch.close()
    for m in members:
        m.close()
    for b in dev + [x for o in outs for x in o]:
        b.free()
