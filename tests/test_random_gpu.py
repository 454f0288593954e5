"""GPU: randomized parity sweep -- random small geometries (odd-ish sizes, padded strides, every resolution scalar the
size allows), random search radius / scalars / iterations / blur radius / levels / blend scalars, every output mode,
single launches and fused periods, against the CPU oracle.  Bar: bit-exact (mode 3 = HSV diagnostic: <= 2 LSB of 8 bits).
Seeds are fixed: a failure names the seed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    hdr = int(rng.integers(0, 2))
    big = seed >= 40                      # a few larger frames: rs up to 3, 16-byte warp threads, several wave tiles per row
    H = int(rng.integers(8, 160) if not big else rng.integers(150, 420)) * 2
    W = int(rng.integers(16, 240) if not big else rng.integers(300, 700)) * 2
    if seed % 3 == 0:                      # exercise the 16-byte fast paths too
        W = (W + 15) // 16 * 16
    si = W + int(rng.choice([0, 0, 2, 6, 16, 34, 1, 7]))      # odd pitches: no vector path applies
    so = W + int(rng.choice([0, 0, 2, 8, 16, 30, 3, 5]))
    max_res = int(rng.choice([270, 270, 64, 48, 1000]))        # forces rs = 0 .. 2 at these sizes
    if big and seed % 2 == 0:
        max_res = 1000                     # grids wider than 512: windows > 32 at the levels that have the neighbour term
    R = int(rng.integers(2, 17))
    delta = int(rng.integers(0, 11))
    nb = int(rng.integers(0, 11))
    iters = int(rng.choice([0, 0, 0, 1, 2, 3, 5]))
    blur = int(rng.choice([4, 4, 4, 1, 2, 7, 12]))
    black = float(rng.choice([0.0, 0.0, 16.0, 3.5]))
    white = float(rng.choice([255.0, 255.0, 235.0, 200.25]))
    return dict(hdr=hdr, H=H, W=W, si=si, so=so, max_res=max_res, R=R, delta=delta, nb=nb, iters=iters, blur=blur,
                black=black, white=white, seed=seed)


@pytest.mark.parametrize("seed", list(range(64)))
def test_random_configuration_matches_oracle(native_lib, seed):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    k = _case(seed)
    rng = np.random.default_rng(5000 + seed)
    hdr, H, W = k["hdr"], k["H"], k["W"]
    sc = synth.Scene(H, W, bool(hdr), seed=300 + seed, in_stride=k["si"], max_rect_speed=int(rng.integers(2, 40)))
    f = [sc.frame(i) for i in range(4)]
    if seed % 5 == 4:     # white noise over the full code range (P010 low bits set, 8-bit 0..255): no structure, worst case for wrap-around
        f = [synth.random_frame(H, W, bool(hdr), seed=7000 + 10 * seed + i, in_stride=k["si"]) for i in range(4)]
    g = oracle.make_geom(hdr, H, W, k["si"], k["so"], k["max_res"])
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    c = cls(H, W, k["si"], k["so"], k["delta"], k["nb"], k["black"], k["white"], k["max_res"], iterations=k["iters"],
            blur_radius=k["blur"], search_radius=k["R"], flags=(capi.HF_FLAG_ASYNC if seed % 2 else 0) | (capi.HF_FLAG_DUAL_STREAM if seed % 6 == 1 else 0))
    for x in f[:3]:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    c.sync()
    off_a, blur_a, tot_a, oob_a = oracle.calculate_optical_flow(f[1], f[2], g, k["R"], k["iters"], k["delta"], k["nb"], k["blur"])
    # oob_a > 0: some sample position left the range the reference defines (its single reflection indexes outside the
    # frame, calcDeltaSumsKernelSDR.h:86-95 -- only with huge scalars / tiny frames).  The oracle and the HIP path both
    # clamp there, so they still have to agree with each other; only the comparison with the reference is off.
    assert (c.readOffsets() == off_a).all(), (k, oob_a)
    assert c.m_totalFrameDelta == tot_a, (k, oob_a)
    assert (c.readBlurredFlow(1) == oracle.blur_flow(c.readOffsets(), g, k["blur"])).all(), k
    c.updateFrame(f[3])
    c.calculateOpticalFlow()
    c.sync()
    flow = c.readBlurredFlow(0)          # previous flow = what warpFrames uses
    dt = np.uint16 if hdr else np.uint8
    n_el = c.output_frame_bytes // np.dtype(dt).itemsize
    ts = [0.0, 1.0] + [float(np.float32(x)) for x in rng.random(3)]
    for mode in range(7):
        for t in ts[:3] if mode != 2 else ts:
            c.warpFrames(t, mode)
            out = c.downloadFrame()
            ref = oracle.warp_frames(f[1], f[2], flow, g, t, mode, k["black"], k["white"])
            if mode == 3:
                tol = 2 * (256 if hdr else 1)
                assert np.abs(out.astype(np.int64) - ref.astype(np.int64)).max() <= tol, (k, mode, t)
            else:
                assert (out == ref).all(), (k, mode, t)
    # the same blend scalars as one fused period launch
    outs = [DeviceBuffer(c.output_frame_bytes) for _ in ts]
    c.interpolateOnly(ts, [o.ptr for o in outs], 2)
    c.sync()
    for t, o in zip(ts, outs):
        ref = oracle.warp_frames(f[1], f[2], flow, g, t, 2, k["black"], k["white"])
        # caller-owned output buffers: only the W valid elements of every row are defined (the stride padding is not written)
        got = o.download(dt, n_el).reshape(H + H // 2, k["so"])[:, :W]
        assert (got == ref.reshape(H + H // 2, k["so"])[:, :W]).all(), (k, "fused", t)
    c.copyFrame()
    assert (c.downloadFrame() == oracle.copy_frame(f[1], g, k["black"], k["white"])).all(), k
    c.close()


@pytest.mark.parametrize("hdr,H,W", [(0, 4, 4), (1, 4, 4), (0, 6, 10), (1, 8, 6), (0, 4, 64), (1, 64, 4), (0, 10, 34), (1, 16, 16), (0, 2 * 135, 2 * 3), (1, 12, 130)])
def test_tiny_frames_match_oracle(native_lib, hdr, H, W):
    """The smallest frames the C ABI accepts (even sizes >= 4): degenerate pyramids (1-3 levels), grids narrower than one
    strip / one tile, all output modes."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    f = [synth.random_frame(H, W, bool(hdr), seed=900 + 7 * i + H + W) for i in range(4)]
    g = oracle.make_geom(hdr, H, W)
    for R in (2, 5, 16):
        c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=R)
        for x in f[:3]:
            c.updateFrame(x)
        c.calculateOpticalFlow()
        off, blur, tot, oob = oracle.calculate_optical_flow(f[1], f[2], g, R)
        # oob > 0 (offsets larger than these tiny frames): undefined in the reference, clamped identically by oracle and HIP
        assert (c.readOffsets() == off).all(), (R, "offsets", oob)
        assert c.m_totalFrameDelta == tot
        assert (c.readBlurredFlow(1) == oracle.blur_flow(c.readOffsets(), g, 4)).all(), (R, "blur")
        c.updateFrame(f[3]); c.calculateOpticalFlow()
        flow = c.readBlurredFlow(0)
        for mode in (0, 1, 2, 4, 5, 6):
            for t in (0.0, 0.43, 1.0):
                c.warpFrames(t, mode)
                assert (c.downloadFrame() == oracle.warp_frames(f[1], f[2], flow, g, t, mode)).all(), (R, mode, t)
        c.copyFrame()
        assert (c.downloadFrame() == oracle.copy_frame(f[1], g)).all()
        c.close()
