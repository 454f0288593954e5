"""GPU: the HIP path against THE REFERENCE ITSELF run live on the same MI355X (oracle/_ref: unmodified
reference host code + OpenCL kernels, driven by ref_runner), on seeds that are not in the golden set.
Skipped when oracle/_ref or an OpenCL GPU is not available."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hdr,H,W,si,so,R,delta,nb,seed,max_res", [
    (0, 270, 480, 0, 0, 16, 8, 6, 501, 270),
    (1, 270, 480, 512, 496, 12, 8, 6, 502, 270),
    (0, 1080, 1920, 0, 0, 16, 8, 6, 503, 270),
    (1, 1080, 1920, 0, 0, 5, 6, 10, 504, 270),
    (1, 2160, 3840, 0, 0, 16, 8, 6, 505, 270),
    (0, 568, 1388, 0, 0, 12, 3, 4, 506, 1000),      # maxCalcRes above the default: 1388-wide grid, windows > 32 at neighbour-term levels
    (1, 540, 960, 0, 0, 9, 8, 6, 507, 540),         # rs = 0 at qHD
])
def test_hip_matches_live_reference(native_lib, hdr, H, W, si, so, R, delta, nb, seed, max_res):
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) or an OpenCL GPU is not available")
    sc = synth.Scene(H, W, bool(hdr), seed, in_stride=si)
    f = [sc.frame(k) for k in range(4)]
    tvals = [0.0, 0.1998, 0.3996, 0.5994, 0.999]
    s = oracle.RefSession(hdr, H, W, si, so, delta, nb, 0.0, 255.0, max_res)
    s.radius(R)
    for x in f[:3]:
        s.update(x)
    s.calc(); s.stats(); s.dump_offsets("off"); s.dump_blurred(1, "blur")
    s.update(f[3]); s.calc(); s.stats()
    for m in (0, 1, 2):
        for t in tvals:
            s.warp(t, m); s.download(f"w{m}_{t}")
    s.copy(); s.download("copy")
    js, ref = s.run()

    c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, si, so, delta, nb, 0.0, 255.0, max_res)
    c.m_opticalFlowSearchRadius = R
    for x in f[:3]:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    assert c.m_totalFrameDelta == js[0]["total_frame_delta"]
    assert (c.readOffsets() == ref["off"]).all()
    assert (c.readBlurredFlow(1) == ref["blur"]).all()
    c.updateFrame(f[3]); c.calculateOpticalFlow()
    assert c.m_totalFrameDelta == js[1]["total_frame_delta"]
    for m in (0, 1, 2):
        for t in tvals:
            c.warpFrames(t, m)
            assert (c.downloadFrame() == ref[f"w{m}_{t}"]).all(), f"mode {m} t {t}"
    c.copyFrame()
    assert (c.downloadFrame() == ref["copy"]).all()
    # the same five outputs as ONE fused period launch (hf_interpolate_period_ex): at 2160p every thread produces all of
    # them (the launch shape bench.py times), judged here by the live reference (warpFrameKernelHDR.h:116-184)
    from hopperrender_amd import capi
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch
    dt = np.uint16 if hdr else np.uint8
    outs = [DeviceBuffer(c.output_frame_bytes) for _ in tvals]
    zeros = np.zeros(c.output_frame_bytes, np.uint8)
    for b in outs:
        b.upload(zeros)            # the padding behind each row (output stride > width) is never written: zero as in the reference's buffer
    for m in (2, 0, 1):
        c.interpolateOnly(tvals, [b.ptr for b in outs], m)
        c.sync()
        for j, t in enumerate(tvals):
            assert (outs[j].download(dt) == ref[f"w{m}_{t}"]).all(), f"fused period, mode {m} t {t}"
    c.close()
    if H * W <= 1920 * 1080 and si == 0:
        # frames up to 1080p produce all outputs per thread only inside a batch of >= 11 members: 12 members, same frames
        n = 12
        cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
        members = [cls(H, W, si, so, delta, nb, 0.0, 255.0, max_res, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
        batch = FlowBatch(members)
        dev = []
        for x in f:
            b = DeviceBuffer(x.nbytes); b.upload(x); dev.append(b)
        for k in range(4):
            batch.updateFramesDeviceRef([dev[k].ptr] * n)
            if k >= 2:
                batch.calculateOpticalFlow()
        mouts = [[DeviceBuffer(len(zeros)) for _ in tvals] for _ in range(n)]
        plans = [tvals[i % 5:] + tvals[:i % 5] for i in range(n)]
        batch.interpolatePeriod(plans, [[b.ptr for b in mo] for mo in mouts], 2)
        for i, mbr in enumerate(members):
            mbr.sync()
            assert mbr.m_totalFrameDelta == js[1]["total_frame_delta"]
            for j, t in enumerate(plans[i]):
                assert (mouts[i][j].download(dt) == ref[f"w2_{t}"]).all(), f"batched fused period, member {i} t {t}"
        batch.close()
        for mbr in members:
            mbr.close()


@pytest.mark.parametrize("hdr,H,W", [(0, 4, 4), (1, 4, 4), (0, 6, 10)])
def test_tiny_frames_match_live_reference(native_lib, hdr, H, W):
    """4-row frames: the chroma plane has 2 rows and mirrorCoordinate's clamp(r, 1, dim - 2) has min > max -- OpenCL's
    min(max()) order decides.  HIP path and oracle against the reference itself."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) or an OpenCL GPU is not available")
    f = [synth.random_frame(H, W, bool(hdr), seed=900 + 7 * i + H + W) for i in range(4)]
    s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, 0.0, 255.0, 270)
    s.radius(5)
    for x in f[:3]:
        s.update(x)
    s.calc(); s.update(f[3]); s.calc()
    s.dump_blurred(0, "flow")
    for m in (0, 1, 2):
        s.warp(0.43, m); s.download(f"w{m}")
    s.copy(); s.download("copy")
    js, ref = s.run()
    c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=5)
    for x in f[:3]:
        c.updateFrame(x)
    c.calculateOpticalFlow(); c.updateFrame(f[3]); c.calculateOpticalFlow()
    assert (c.readBlurredFlow(0) == ref["flow"]).all()
    g = oracle.make_geom(hdr, H, W)
    for m in (0, 1, 2):
        c.warpFrames(0.43, m)
        assert (c.downloadFrame() == ref[f"w{m}"]).all(), m
        assert (oracle.warp_frames(f[1], f[2], ref["flow"], g, 0.43, m) == ref[f"w{m}"]).all(), ("oracle", m)
    c.copyFrame()
    assert (c.downloadFrame() == ref["copy"]).all()
    c.close()


@pytest.mark.parametrize("hdr", [0, 1])
def test_degenerate_levels_match_live_reference(native_lib, hdr):
    """Levels a settings UI can produce but no sane user wants: white == black, white = 0, black > white, out-of-range
    values (division by zero / negative scale inside apply_levels): blend and copy against the reference itself."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) or an OpenCL GPU is not available")
    H, W = 36, 64
    sc = synth.Scene(H, W, bool(hdr), 5)
    f = [sc.frame(k) for k in range(4)]
    g = oracle.make_geom(hdr, H, W)
    for bk, wh in [(16.0, 16.0), (0.0, 0.0), (200.0, 100.0), (-20.0, 300.0), (0.0, 1e-3), (255.0, 0.0)]:
        s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, bk, wh, 270)
        s.radius(8)
        for x in f[:3]:
            s.update(x)
        s.calc(); s.update(f[3]); s.calc()
        s.warp(0.4, 2); s.download("w2")
        s.copy(); s.download("copy")
        js, ref = s.run()
        c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, 0, 0, 8, 6, bk, wh, 270, search_radius=8)
        for x in f[:3]:
            c.updateFrame(x)
        c.calculateOpticalFlow(); c.updateFrame(f[3]); c.calculateOpticalFlow()
        c.warpFrames(0.4, 2)
        assert (c.downloadFrame() == ref["w2"]).all(), (bk, wh)
        assert (oracle.warp_frames(f[1], f[2], c.readBlurredFlow(0), g, 0.4, 2, bk, wh) == ref["w2"]).all(), ("oracle", bk, wh)
        c.copyFrame()
        assert (c.downloadFrame() == ref["copy"]).all(), (bk, wh)
        assert (oracle.copy_frame(f[1], g, bk, wh) == ref["copy"]).all(), ("oracle", bk, wh)
        c.close()


@pytest.mark.parametrize("seed", list(range(52)))
def test_random_configurations_match_live_reference(native_lib, seed):
    """The randomized sweep of test_random_gpu.py, but judged by the reference itself (oracle/_ref on this GPU) instead
    of the oracle: geometry, pitches, resolution scalar, search radius 2..16, scalars, levels, all production modes.
    The oracle is checked against the same outputs, so every seed pins it once more."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) or an OpenCL GPU is not available")
    rng = np.random.default_rng(9000 + seed)
    hdr = int(rng.integers(0, 2))
    H = int(rng.integers(8, 120)) * 2
    W = int(rng.integers(16, 200)) * 2
    if seed >= 40:                          # frames up to ~1100 x 2000: resolution scalars 2-3, 16-byte warp threads, many tiles
        H = int(rng.integers(280, 560)) * 2
        W = int(rng.integers(500, 1000)) * 2
        if seed % 2:
            W = (W + 15) // 16 * 16
    si = W + int(rng.choice([0, 0, 2, 16, 6]))
    so = W + int(rng.choice([0, 0, 2, 16, 10]))
    max_res = int(rng.choice([270, 270, 64, 40, 1000]))
    R = int(rng.integers(2, 17))
    delta, nb = int(rng.integers(0, 11)), int(rng.integers(0, 11))
    bk, wh = float(rng.choice([0.0, 16.0, 3.5])), float(rng.choice([255.0, 235.0, 200.25]))
    sc = synth.Scene(H, W, bool(hdr), seed=700 + seed, in_stride=si, max_rect_speed=int(rng.integers(2, 30)))
    f = [sc.frame(k) for k in range(4)]
    ts = [0.0, float(np.float32(rng.random())), 0.999]
    s = oracle.RefSession(hdr, H, W, si, so, delta, nb, bk, wh, max_res)
    s.radius(R)
    for x in f[:3]:
        s.update(x)
    s.calc(); s.stats(); s.dump_offsets("off"); s.dump_blurred(1, "blur")
    s.update(f[3]); s.radius(R); s.calc(); s.stats()
    for m in (0, 1, 2):
        for t in ts:
            s.warp(t, m); s.download(f"w{m}_{t}")
    s.copy(); s.download("copy")
    js, ref = s.run()
    g = oracle.make_geom(hdr, H, W, si, so, max_res)
    off_o, blur_o, tot_o, oob = oracle.calculate_optical_flow(f[1], f[2], g, R, 0, delta, nb, 4)
    c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, si, so, delta, nb, bk, wh, max_res, search_radius=R)
    for x in f[:3]:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    cfg = dict(hdr=hdr, H=H, W=W, si=si, so=so, max_res=max_res, R=R, delta=delta, nb=nb, bk=bk, wh=wh)
    if oob == 0:    # otherwise the reference reads outside its frame buffers (undefined)
        assert c.m_totalFrameDelta == js[0]["total_frame_delta"] == tot_o, cfg
        assert (c.readOffsets() == ref["off"]).all() and (off_o == ref["off"]).all(), cfg
        assert (c.readBlurredFlow(1) == ref["blur"]).all() and (blur_o == ref["blur"]).all(), cfg
        c.updateFrame(f[3]); c.calculateOpticalFlow()
        for m in (0, 1, 2):
            for t in ts:
                c.warpFrames(t, m)
                assert (c.downloadFrame() == ref[f"w{m}_{t}"]).all(), (cfg, m, t)
                assert (oracle.warp_frames(f[1], f[2], ref["blur"], g, t, m, bk, wh) == ref[f"w{m}_{t}"]).all(), (cfg, "oracle", m, t)
        c.copyFrame()
        assert (c.downloadFrame() == ref["copy"]).all(), cfg
    c.close()
