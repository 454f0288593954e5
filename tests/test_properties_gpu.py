"""GPU: size-independent properties at BASELINE.json's full sizes (1080p SDR, 2160p HDR), where running the
CPU oracle for every case would take too long (the live-reference test covers one full-size case each)."""
import numpy as np
import pytest

from helpers import sha

pytestmark = pytest.mark.gpu

FULL = [(0, 1080, 1920), (1, 2160, 3840)]


def calc_for(hdr, H, W, **kw):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    return (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, **kw)


@pytest.mark.parametrize("hdr,H,W", FULL)
def test_identical_frames_give_zero_flow_and_identity_gather(native_lib, hdr, H, W):
    from hopperrender_amd import synth
    f = synth.Scene(H, W, bool(hdr), 7).frame(0)
    c = calc_for(hdr, H, W, search_radius=16)
    for _ in range(3):
        c.updateFrame(f)
    c.calculateOpticalFlow()
    c.calculateOpticalFlow()
    assert not c.readOffsets().any() and not c.readBlurredFlow(0).any()
    # zero flow: gather modes return the source with the reference's [1, dim-2] edge clamp (warpFrameKernelSDR.h:12-20)
    S = W
    y = f[:H * S].reshape(H, S)
    uv = f[H * S:].reshape(H // 2, S)
    def mw(dim):   # mirrorCoordinate of the warp kernel: 0 -> 1, dim-1 -> dim-3, identity on [1, dim-2]
        p = np.arange(dim)
        r = np.where(p >= dim - 1, 2 * (dim - 2) - p, np.where(p < 1, 1 - p, p))
        return np.clip(r, 1, dim - 2)
    ey = y[mw(H)][:, mw(W)]
    xs = mw(W)
    euv = uv[mw(H // 2)][:, (xs & ~1) + (np.arange(W) & 1)]
    want = np.concatenate([ey.reshape(-1), euv.reshape(-1)])
    for mode in (0, 1):
        c.warpFrames(0.37, mode)
        assert (c.downloadFrame() == want).all()
    c.close()


@pytest.mark.parametrize("hdr,H,W", FULL)
def test_copy_levels_and_determinism(native_lib, hdr, H, W):
    from hopperrender_amd import capi, synth
    sc = synth.Scene(H, W, bool(hdr), 8)
    fr = [sc.frame(k) for k in range(4)]
    hashes = []
    for flags in (0, capi.HF_FLAG_NO_GRAPH | capi.HF_FLAG_NO_LAZY_ARGMIN, capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM):
        c = calc_for(hdr, H, W, search_radius=16, flags=flags)
        for f in fr[:3]:
            c.updateFrame(f)
        c.calculateOpticalFlow()
        c.updateFrame(fr[3])
        c.calculateOpticalFlow()
        h = [sha(c.readOffsets()), sha(c.readBlurredFlow(0)), str(c.m_totalFrameDelta)]
        for t in (0.1998, 0.7992):
            c.warpFrames(t, 2)
            h.append(sha(c.downloadFrame()))
        c.copyFrame()
        out = c.downloadFrame()
        h.append(sha(out))
        # default levels are NOT the identity in the reference as built for gfx950: v * rcp(255) * 255 lands one
        # code low for some v (SDR), and HDR stretches 0..65280 to 0..65535 (opticalFlowCalcHDR.cpp:173-174)
        from oracle import oracle
        assert (out == oracle.copy_frame(fr[1], oracle.make_geom(hdr, H, W))).all()
        d = out.astype(np.int64) - fr[1].astype(np.int64)
        if not hdr:
            assert d.min() >= -1 and d.max() <= 0
        hashes.append(h)
        c.close()
    assert hashes[0] == hashes[1] == hashes[2]   # a checksum of checksums: graph / eager / lazy / dual-stream agree


def test_blur_radius_extension_matches_oracle_full_grid(native_lib):
    """BASELINE config 5: large blur radius at the 480x270 grid of 2160p."""
    from hopperrender_amd import synth
    from oracle import oracle
    H, W = 2160, 3840
    sc = synth.Scene(H, W, True, 9)
    f = [sc.frame(k) for k in range(3)]
    g = oracle.make_geom(1, H, W)
    ref_off = None
    for r in (16, 32, 64):   # 64: the largest the C ABI accepts (101 KB of LDS per workgroup)
        c = calc_for(1, H, W, search_radius=16, blur_radius=r)
        c.m_neighborBiasScalar = 10
        for x in f:
            c.updateFrame(x)
        c.calculateOpticalFlow()
        off = c.readOffsets()
        if ref_off is None:
            ref_off = off
        assert (off == ref_off).all()
        assert (c.readBlurredFlow(1) == oracle.blur_flow(off, g, r)).all()
        c.close()


def test_cli_interpolates_a_raw_clip(native_lib, tmp_path):
    """hopperrender_amd.cli = raw NV12 in/out around the filter-protocol replay."""
    from hopperrender_amd import cli, synth
    from hopperrender_amd.protocol import SOURCE_24, TARGET_60, BlendSchedule
    H, W, n = 180, 320, 5
    sc = synth.Scene(H, W, False, 21)
    clip = np.concatenate([sc.frame(k) for k in range(n)])
    src, dst = tmp_path / "in.nv12", tmp_path / "out.nv12"
    clip.tofile(str(src))
    cli.main([str(src), str(dst), "--width", str(W), "--height", str(H), "--radius", "8"])
    out = np.fromfile(str(dst), dtype=np.uint8)
    n_out = sum(len(p) for p in BlendSchedule(SOURCE_24, TARGET_60).plan(n))
    assert out.size == n_out * (H * W * 3 // 2)
    from oracle import oracle
    assert (out[:H * W * 3 // 2] == oracle.copy_frame(sc.frame(0), oracle.make_geom(0, H, W))).all()   # first period: copy


@pytest.mark.parametrize("hdr", [False, True])
def test_cli_y4m_matches_raw(native_lib, tmp_path, hdr):
    """A .y4m clip (planar 4:2:0, 8 / 10 bit) gives exactly the frames of the same clip fed as raw NV12 / P010."""
    import io
    from hopperrender_amd import cli, synth, y4m
    H, W, n = 180, 320, 5
    sc = synth.Scene(H, W, hdr, 33)
    frames = [sc.frame(k) for k in range(n)]
    raw_in, raw_out = tmp_path / "in.raw", tmp_path / "out.raw"
    np.concatenate(frames).tofile(str(raw_in))
    y_in, y_out = tmp_path / "in.y4m", tmp_path / "out.y4m"
    with open(y_in, "wb") as f:
        w = y4m.Y4MWriter(f, W, H, 24000, 1001, hdr)
        for fr in frames:
            w.write(fr)
    cli.main([str(raw_in), str(raw_out), "--width", str(W), "--height", str(H), "--radius", "8"] + (["--hdr"] if hdr else []))
    cli.main([str(y_in), str(y_out), "--radius", "8"])
    dt = np.uint16 if hdr else np.uint8
    raw = np.fromfile(str(raw_out), dtype=dt).reshape(-1, H * W * 3 // 2)
    with open(y_out, "rb") as f:
        r = y4m.Y4MReader(f)
        assert (r.width, r.height, r.hdr, r.fps_num, r.fps_den) == (W, H, hdr, 60, 1)
        got = list(r)
    assert len(got) == raw.shape[0] > n
    for a, b in zip(got, raw):
        # Y4M keeps the 10-bit code only: P010's low 6 bits (levels stretch residue) are cut on write
        assert (a == ((b >> 6) << 6 if hdr else b)).all()


@pytest.mark.parametrize("hdr", [0, 1])
def test_long_run_last_period_matches_oracle(native_lib, hdr):
    """60 source periods through hf_interpolate_period (graph replay, ring / ping-pong phases cycling, fused warps into
    caller buffers): the outputs of the LAST period still equal the oracle's, which only needs the last three frames."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule
    from oracle import oracle
    H, W, n = 180, 320, 60
    sc = synth.Scene(H, W, bool(hdr), 61)
    frames = [sc.frame(k % 9) for k in range(n)]
    dev = []
    for f in frames[:9]:
        b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    c = calc_for(hdr, H, W, search_radius=11, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING)
    plan = BlendSchedule(SOURCE_24, 83333).plan(n)
    outs = [DeviceBuffer(c.output_frame_bytes) for _ in range(6)]
    for k in range(n):
        c.interpolatePeriod(dev[k % 9].ptr, plan[k], [o.ptr for o in outs], 2)
    c.sync()
    g = oracle.make_geom(hdr, H, W)
    f0, f1, f2 = frames[n - 3], frames[n - 2], frames[n - 1]
    _, blur_prev, _, _ = oracle.calculate_optical_flow(f0, f1, g, 11)       # flow the last period's warps use
    off, blur, tot, _ = oracle.calculate_optical_flow(f1, f2, g, 11)
    assert (c.readOffsets() == off).all() and (c.readBlurredFlow(1) == blur).all() and c.m_totalFrameDelta == tot
    assert (c.readBlurredFlow(0) == blur_prev).all()
    dt = np.uint16 if hdr else np.uint8
    for t, o in zip(plan[n - 1], outs):
        assert (o.download(dt, H * W * 3 // 2) == oracle.warp_frames(f0, f1, blur_prev, g, t, 2)).all(), t
    c.close()


def test_period_with_many_outputs_and_edge_scalars(native_lib):
    """hf_interpolate_period with more outputs than one fused launch takes (6), t = 0 and t = 1 exactly, duplicates,
    and an empty period; every output equals the single-launch warpFrames result."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer
    H, W, hdr = 180, 320, 1
    sc = synth.Scene(H, W, True, 73)
    f = [sc.frame(k) for k in range(4)]
    dev = [DeviceBuffer(x.nbytes) for x in f]
    for d, x in zip(dev, f):
        d.upload(x)
    c = calc_for(hdr, H, W, search_radius=9, flags=capi.HF_FLAG_ASYNC)
    for d in dev[:3]:
        c.updateFrameDeviceRef(d.ptr)
    c.calculateOpticalFlow()
    ts = [0.0, 1.0, 0.5, 0.5, 0.125, 0.999, 0.001, 0.25, 0.75, 1.0, 0.0, 0.3996, 0.7992]      # 13 outputs: 6 + 6 + 1
    outs = [DeviceBuffer(c.output_frame_bytes) for _ in ts]
    c.interpolatePeriod(dev[3].ptr, ts, [o.ptr for o in outs], 2)
    c.interpolatePeriod(0, [], [], 2)             # no new frame, no outputs: just another flow calculation
    c.sync()
    ref = calc_for(hdr, H, W, search_radius=9)
    for x in f:
        ref.updateFrame(x)
        if x is f[2] or x is f[3]:
            ref.calculateOpticalFlow()
    for t, o in zip(ts, outs):
        ref.warpFrames(t, 2)
        assert (o.download(np.uint16, H * W * 3 // 2) == ref.downloadFrame()).all(), t
    c.close(); ref.close()


@pytest.mark.parametrize("hdr", [0, 1])
def test_misaligned_device_pointers_take_the_generic_paths(native_lib, hdr):
    """Source frames and output buffers that are only element-aligned (not 4 / 16 bytes): none of the vector paths may be
    selected, the results stay identical."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer
    H, W = 180, 320
    el = 2 if hdr else 1
    sc = synth.Scene(H, W, bool(hdr), 91)
    f = [sc.frame(k) for k in range(4)]
    off = el                                     # one element off every alignment
    dev = [DeviceBuffer(x.nbytes + 64) for x in f]
    lib = capi.load()
    for d, x in zip(dev, f):
        assert lib.hf_memcpy_h2d(0, d.ptr + off, x.ctypes.data, x.nbytes) == 0
    c = calc_for(hdr, H, W, search_radius=9, flags=capi.HF_FLAG_ASYNC)
    ref = calc_for(hdr, H, W, search_radius=9)
    for d, x in zip(dev[:3], f[:3]):
        c.updateFrameDeviceRef(d.ptr + off)
        ref.updateFrame(x)
    c.calculateOpticalFlow(); ref.calculateOpticalFlow()
    ts = [0.0, 0.3996, 0.7992, 1.0]
    outs = [DeviceBuffer(c.output_frame_bytes + 64) for _ in ts]
    c.interpolatePeriod(dev[3].ptr + off, ts, [o.ptr + off for o in outs], 2)
    c.sync()
    ref.updateFrame(f[3]); ref.calculateOpticalFlow()
    assert (c.readOffsets() == ref.readOffsets()).all() and (c.readBlurredFlow(1) == ref.readBlurredFlow(1)).all()
    dt = np.uint16 if hdr else np.uint8
    n = H * W * 3 // 2
    for t, o in zip(ts, outs):
        ref.warpFrames(t, 2)
        got = np.empty(n, dtype=dt)
        assert lib.hf_memcpy_d2h(0, got.ctypes.data, o.ptr + off, got.nbytes) == 0
        assert (got == ref.downloadFrame()).all(), t
    c.close(); ref.close()
