"""GPU: hf_batch (calculateOpticalFlow of several contexts as one set of launches, include/hopperflow.h) must give
every member exactly what its own hf_calculate_optical_flow gives -- and through it the reference's results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def calc_for(hdr, H, W, dual=False, **kw):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    flags = capi.HF_FLAG_ASYNC | (capi.HF_FLAG_DUAL_STREAM if dual else 0)
    return (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, flags=flags, **kw)


@pytest.mark.parametrize("hdr,H,W,n,R,dual", [(0, 360, 640, 3, 8, False), (1, 360, 640, 8, 16, False), (0, 1080, 1920, 4, 16, False),
                                             (1, 2160, 3840, 2, 5, False), (0, 274, 486, 5, 16, False),
                                             (1, 360, 640, 5, 16, True), (0, 1080, 1920, 4, 9, True)])
def test_batch_equals_single_contexts(native_lib, hdr, H, W, n, R, dual):
    """dual: HF_FLAG_DUAL_STREAM members -- their warps go to the batch's shared warp streams and overlap the chain."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch
    scenes = [synth.Scene(H, W, bool(hdr), 100 + 7 * i) for i in range(n)]
    frames = [[sc.frame(k) for k in range(6)] for sc in scenes]
    singles = [calc_for(hdr, H, W, search_radius=R) for _ in range(n)]
    members = [calc_for(hdr, H, W, dual=dual, search_radius=R) for _ in range(n)]
    batch = FlowBatch(members)
    assert len(batch) == n
    ts = [0.0, 0.3996, 0.7992]
    outs_s = [[DeviceBuffer(singles[0].output_frame_bytes) for _ in ts] for _ in range(n)]
    outs_b = [[DeviceBuffer(singles[0].output_frame_bytes) for _ in ts] for _ in range(n)]
    for i in range(n):
        for k in range(2):
            singles[i].updateFrame(frames[i][k])
            members[i].updateFrame(frames[i][k])
    for k in range(2, 6):            # four rounds: every ring phase / flow ping-pong phase, graph capture and replay
        for i in range(n):
            singles[i].updateFrame(frames[i][k])
            singles[i].calculateOpticalFlow()
            members[i].updateFrame(frames[i][k])
        batch.calculateOpticalFlow()
        for i in range(n):
            singles[i].interpolateOnly(ts, [b.ptr for b in outs_s[i]], 2)
            members[i].interpolateOnly(ts, [b.ptr for b in outs_b[i]], 2)
        for i in range(n):
            singles[i].sync()
            members[i].sync()
            assert members[i].m_totalFrameDelta == singles[i].m_totalFrameDelta, (k, i)
            assert (members[i].readOffsets() == singles[i].readOffsets()).all(), (k, i)
            assert (members[i].readBlurredFlow(1) == singles[i].readBlurredFlow(1)).all(), (k, i)
            assert (members[i].readBlurredFlow(0) == singles[i].readBlurredFlow(0)).all(), (k, i)
            dt = np.uint16 if hdr else np.uint8
            for a, b in zip(outs_s[i], outs_b[i]):
                assert (a.download(dt) == b.download(dt)).all(), (k, i)
    # distinct scenes must give distinct flows (the members are not aliased to one pair)
    assert any((members[0].readBlurredFlow(1) != members[i].readBlurredFlow(1)).any() for i in range(1, n))
    batch.close()
    # after the batch is gone the members are ordinary contexts again
    members[0].updateFrame(frames[0][0]); members[0].calculateOpticalFlow()
    singles[0].updateFrame(frames[0][0]); singles[0].calculateOpticalFlow()
    members[0].sync(); singles[0].sync()
    assert (members[0].readBlurredFlow(1) == singles[0].readBlurredFlow(1)).all()
    for c in singles + members:
        c.close()


def test_batch_matches_reference_golden(native_lib):
    """Members fed with the golden case's frames reproduce the reference's flow (pins the batch to the reference)."""
    from helpers import ALL_GOLDEN, Golden
    from hopperrender_amd import capi
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    name = [n for n in ALL_GOLDEN if "180" in n or "360" in n][0]
    g = Golden(name)
    frames = g.frames()
    key = g.keys[0]
    R, delta, nb = g.params(key)
    cls = OpticalFlowCalcHDR if g.case["hdr"] else OpticalFlowCalcSDR
    ms = [cls(g.case["H"], g.case["W"], g.case["si"], g.case["so"], delta, nb, 0.0, 255.0, 270, flags=capi.HF_FLAG_ASYNC, search_radius=R)
          for _ in range(3)]
    b = FlowBatch(ms)
    for m in ms:
        for f in frames[:3]:
            m.updateFrame(f)
    b.calculateOpticalFlow()
    for m in ms:
        m.sync()
        assert m.m_totalFrameDelta == g.meta[key]["stats_a"]["total_frame_delta"]
        assert (m.readOffsets() == g.arr(key, "off_a")).all()
        assert (m.readBlurredFlow(1) == g.arr(key, "blur_a")).all()
    b.close()
    for m in ms:
        m.close()


def test_batch_rejects_incompatible_members(native_lib):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR
    a = calc_for(0, 360, 640)
    b = calc_for(0, 180, 320)
    with pytest.raises(capi.HopperFlowError):
        FlowBatch([a, b])
    with pytest.raises(capi.HopperFlowError):
        FlowBatch([a, a])
    sync_ctx = OpticalFlowCalcSDR(360, 640, 0, 0, 8, 6, 0.0, 255.0, 270)   # blocking context: not batchable
    with pytest.raises(capi.HopperFlowError):
        FlowBatch([a, sync_ctx])
    c = calc_for(0, 360, 640)
    bt = FlowBatch([a, c])
    c.m_opticalFlowSearchRadius = 9          # members must agree on the parameters at call time
    with pytest.raises(capi.HopperFlowError):
        bt.calculateOpticalFlow()
    bt.close()
    for x in (a, b, c, sync_ctx):
        x.close()


def test_batch_members_out_of_step_and_parameter_churn(native_lib):
    """Members whose rings / flow ping-pong are in different phases (one joined a frame later), and more
    (R, delta, neighbour) combinations than the graph caches hold (96): every result still equals the single-context one."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import FlowBatch
    hdr, H, W, n = 0, 180, 320, 3
    scenes = [synth.Scene(H, W, False, 900 + i) for i in range(n)]
    frames = [[sc.frame(k) for k in range(8)] for sc in scenes]
    singles = [calc_for(hdr, H, W, search_radius=8) for _ in range(n)]
    members = [calc_for(hdr, H, W, search_radius=8) for _ in range(n)]
    for i in range(n):
        for k in range(2 + i):                 # member i is i frames (and i flow calculations) ahead
            for c in (singles[i], members[i]):
                c.updateFrame(frames[i][k])
                if k >= 2:
                    c.calculateOpticalFlow()
    batch = FlowBatch(members)
    rng = np.random.default_rng(3)
    combos = [(int(rng.integers(2, 17)), int(rng.integers(0, 11)), int(rng.integers(0, 11))) for _ in range(110)]
    for j, (R, delta, nb) in enumerate(combos):
        new_frame = j % 4 == 0
        for i in range(n):
            for c in (singles[i], members[i]):
                c.m_opticalFlowSearchRadius = R
                c.m_deltaScalar = delta
                c.m_neighborBiasScalar = nb
                if new_frame:
                    c.updateFrame(frames[i][(2 + i + j // 4) % 8])
            singles[i].calculateOpticalFlow()
        batch.calculateOpticalFlow()
        if j % 10 == 9 or j == len(combos) - 1:
            for i in range(n):
                singles[i].sync(); members[i].sync()
                assert members[i].m_totalFrameDelta == singles[i].m_totalFrameDelta, (j, i)
                assert (members[i].readOffsets() == singles[i].readOffsets()).all(), (j, i)
                assert (members[i].readBlurredFlow(1) == singles[i].readBlurredFlow(1)).all(), (j, i)
                assert (members[i].readBlurredFlow(0) == singles[i].readBlurredFlow(0)).all(), (j, i)
    batch.close()
    for c in singles + members:
        c.close()


def test_c_batch_driver_example_matches_python_path(native_lib, tmp_path):
    """examples/hf_batch_driver.c (plain C against include/hopperflow.h: contexts, hf_batch, device frames, fused
    periods) prints one checksum per output frame; the same clips through single Python contexts give the same ones."""
    import os
    import subprocess
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "hf_batch_driver"
    lib = os.path.join(root, "hopperrender_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "hf_batch_driver.c"), "-L", lib, "-lhopperflow",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    periods = 4
    out = subprocess.run([str(exe), str(periods)], capture_output=True, text=True, timeout=300, check=True).stdout
    got = {}
    for line in out.strip().splitlines():
        w = line.split()
        got[(int(w[1]), int(w[3]), int(w[5]))] = (int(w[7]), w[9])
    assert len(got) == periods * 2 * 3

    H, W = 180, 320

    def make_frame(clip, k):
        y, x = np.mgrid[0:H, 0:W]
        luma = (((x + 3 * k + 5 * clip) * 7 + (y + 2 * k) * 13 + (((x + 3 * k) >> 4) ^ ((y + 2 * k) >> 4)) * 29) & 0xFF).astype(np.uint8)
        yc, xc = np.mgrid[0:H // 2, 0:W]
        chroma = (128 + (((xc >> 1) + k + clip) * 3 + yc * 5) % 64 - 32).astype(np.uint8)
        return np.concatenate([luma.reshape(-1), chroma.reshape(-1)])

    def fnv1a(a):
        h = 1469598103934665603
        for b in a.tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h

    for clip in range(2):
        c = OpticalFlowCalcSDR(H, W, search_radius=12)
        for k in range(2):
            c.updateFrame(make_frame(clip, k))
        for p in range(periods):
            c.updateFrame(make_frame(clip, p + 2))
            c.calculateOpticalFlow()
            for i, t in enumerate((0.0, 0.3996, 0.7992)):
                c.warpFrames(t, 2)
                assert got[(p, clip, i)] == (c.m_totalFrameDelta, fnv1a(c.downloadFrame())), (p, clip, i)
        c.close()
