"""GPU parity: the HIP path (through the C ABI) against (a) the committed golden vectors produced by
the reference itself and (b) the CPU oracle on seeded inputs.  Bar: bit-exact for the integer flow
stages, gather modes and copy; bit-exact for blend/levels too (the HIP kernels reproduce the fp32
operations of the reference's gfx950 OpenCL build); HSV visualisation (mode 3, atan2/fmod) <= 2 LSB
of 8 bits (<= 2*256 codes in HDR)."""
import numpy as np
import pytest

from helpers import ALL_GOLDEN, Golden, parse_frame_name, sha

pytestmark = pytest.mark.gpu


def make_calc(case, R, delta, nb, **kw):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    cls = OpticalFlowCalcHDR if case["hdr"] else OpticalFlowCalcSDR
    c = cls(case["H"], case["W"], case["si"], case["so"], delta, nb, 0.0, 255.0, case.get("max_res", 270), **kw)
    c.m_opticalFlowSearchRadius = R
    return c


@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_flow_and_frames_match_reference_golden(native_lib, name):
    g = Golden(name)
    frames = g.frames()
    for key in g.keys:
        R, delta, nb = g.params(key)
        c = make_calc(g.case, R, delta, nb)
        for f in frames[:3]:
            c.updateFrame(f)
        c.calculateOpticalFlow()
        assert c.m_totalFrameDelta == g.meta[key]["stats_a"]["total_frame_delta"]
        assert (c.readOffsets() == g.arr(key, "off_a")).all(), f"{name} {key}: raw flow differs"
        assert (c.readBlurredFlow(1) == g.arr(key, "blur_a")).all(), f"{name} {key}: blurred flow differs"
        c.updateFrame(frames[3])
        c.calculateOpticalFlow()
        assert c.m_totalFrameDelta == g.meta[key]["stats_b"]["total_frame_delta"]
        assert (c.readOffsets() == g.arr(key, "off_b")).all()
        assert (c.readBlurredFlow(1) == g.arr(key, "blur_b")).all()
        assert (c.readBlurredFlow(0) == g.arr(key, "blur_a")).all(), "ping-pong: [0] must hold the previous flow"
        for fname in g.frame_names(key):
            kind, mode, t, (bk, wh) = parse_frame_name(fname)
            c.m_outputBlackLevel, c.m_outputWhiteLevel = bk, wh
            if kind == "warp":
                c.warpFrames(t, mode)
            else:
                c.copyFrame()
            out = c.downloadFrame()
            if mode == 3:  # diagnostic HSV (atan2/fmod): tolerance instead of hash
                if g.has(key, fname):
                    d = np.abs(out.astype(np.int64) - g.arr(key, fname).astype(np.int64))
                    assert d.max() <= (2 * 256 if g.case["hdr"] else 2), f"{name} {key} {fname}: max diff {d.max()}"
                continue
            assert sha(out) == g.frame_sha(key, fname), f"{name} {key} {fname}: output frame differs from the reference"
        c.close()


def test_eager_and_graph_paths_agree(native_lib):
    from hopperrender_amd import capi
    g = Golden("sdr_360p")
    frames = g.frames()
    outs = []
    for flags in (0, capi.HF_FLAG_NO_GRAPH, capi.HF_FLAG_ASYNC, capi.HF_FLAG_NO_LAZY_ARGMIN,
                  capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM):
        c = make_calc(g.case, 16, 8, 6, flags=flags)
        for f in frames[:3]:
            c.updateFrame(f)
        c.calculateOpticalFlow()
        c.calculateOpticalFlow()  # second call replays the cached graph
        c.sync()
        outs.append((c.readOffsets(), c.readBlurredFlow(1), c.m_totalFrameDelta))
        c.close()
    for o in outs[1:]:
        assert (o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all() and o[2] == outs[0][2]
    assert (outs[0][0] == g.arr("R16_d8_n6", "off_a")).all()


@pytest.mark.parametrize("hdr,H,W,si,so,R,it,blur", [
    (0, 36, 64, 0, 0, 16, 0, 4),       # tiny: first window 32
    (0, 18, 32, 0, 0, 5, 0, 4),        # tiny: first window 16 (single-launch path from step 0)
    (1, 90, 160, 176, 168, 11, 3, 4),  # runtime `iterations` extension (cfg 1 of BASELINE.json: 3 levels)
    (0, 360, 640, 0, 0, 16, 3, 4),     # BASELINE config 1: 640x360 SDR, 3-level
    (1, 360, 640, 0, 0, 16, 0, 16),    # blur radius extension (BASELINE config 5)
    (0, 270, 480, 0, 0, 13, 0, 32),
    (0, 270, 480, 0, 0, 16, 0, 64),    # the largest radius: window-sum form (even radii), 80 x 80 windows per tile
    (1, 180, 320, 0, 0, 16, 0, 2),     # the smallest even radius
    (0, 270, 480, 0, 0, 9, 0, 7),      # odd radius: pixel form with running row sums
    (0, 270, 482, 0, 0, 16, 0, 12),    # grid of odd width (241): pixel form at an even radius
    (0, 1090, 1922, 1984, 1936, 16, 0, 4),  # ragged 1080p-class, rs=3
    (0, 1080, 1920, 0, 0, 16, 0, 4),        # BASELINE config 2 geometry
    (1, 2160, 3840, 0, 0, 16, 0, 4),        # BASELINE config 3 geometry
])
def test_flow_matches_oracle_on_seeded_inputs(native_lib, hdr, H, W, si, so, R, it, blur):
    from hopperrender_amd import synth
    from oracle import oracle
    case = dict(hdr=hdr, H=H, W=W, si=si, so=so)
    sc = synth.Scene(H, W, bool(hdr), seed=77 + H, in_stride=si)
    f = [sc.frame(k) for k in range(3)]
    g = oracle.make_geom(hdr, H, W, si, so)
    off_o, blur_o, tot_o, oob = oracle.calculate_optical_flow(f[1], f[2], g, R, it, 7, 5, blur)
    c = make_calc(case, R, 7, 5, iterations=it, blur_radius=blur)
    for x in f:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    # (oob > 0 = outside the reference's defined behaviour; the oracle and the HIP path clamp identically there)
    assert (c.readOffsets() == off_o).all(), oob
    assert (c.readBlurredFlow(1) == blur_o).all()
    assert c.m_totalFrameDelta == tot_o
    c.calculateOpticalFlow()
    flow = c.readBlurredFlow(0)
    for t in (0.0, 0.37, 1.0):
        for mode in (0, 1, 2, 3, 4, 5, 6):
            c.warpFrames(t, mode)
            out = c.downloadFrame()
            ref = oracle.warp_frames(f[0], f[1], flow, g, t, mode)
            if mode == 3:   # diagnostic HSV visualisation: atan2 / fmod of the device vs libm, <= 2 LSB of 8 bits
                d = np.abs(out.astype(np.int64) - ref.astype(np.int64))
                assert d.max() <= (2 * 256 if hdr else 2), f"warp t={t} mode=3: max diff {d.max()}"
                continue
            assert (out == ref).all(), f"warp t={t} mode={mode}"
    c.copyFrame()
    assert (c.downloadFrame() == oracle.copy_frame(f[0], g)).all()
    c.close()


def test_levels_match_reference_ramp(native_lib):
    """Every code value through copyFrame at several level settings, against the reference's output."""
    import json
    import os
    from helpers import GOLDEN_DIR
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    z = np.load(os.path.join(GOLDEN_DIR, "levels_ramp.npz"))
    H = W = 256
    for name in z.files:
        if name == "meta":
            continue
        kind, ph, b, w = name.split("_")
        hdr, phase, bk, wh = kind == "hdr", int(ph[1:]), float(b[1:]), float(w[1:])
        n = (H + H // 2) * W
        a = np.arange(n, dtype=np.uint32)
        a[H * W:] += phase * (n - H * W)
        f = (a % 65536).astype(np.uint16) if hdr else (a % 256).astype(np.uint8)
        c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, 0, 0, 8, 6, bk, wh, 270)
        c.updateFrame(f)
        c.copyFrame()
        assert (c.downloadFrame() == z[name]).all(), name
        c.close()


def test_error_behaviour(native_lib):
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    from hopperrender_amd.capi import HopperFlowError
    c = OpticalFlowCalcSDR(180, 320)
    with pytest.raises(HopperFlowError, match="greater than 1.0"):  # opticalFlowCalcSDR.cpp:143-146
        c.warpFrames(1.5, 2)
    with pytest.raises(HopperFlowError):
        OpticalFlowCalcSDR(181, 320)
    assert c.m_opticalFlowSearchRadius == 5  # MIN_SEARCH_RADIUS (opticalFlowCalcSDR.cpp:216)
    assert c.m_frameCount == 0
    c.updateFrame(np.zeros(180 * 320 * 3 // 2, np.uint8))
    assert c.m_frameCount == 1
    c.m_frameCount = 0  # NewSegment (HopperRender.cpp:840)
    assert c.m_frameCount == 0
    c.close()


def test_batched_async_contexts_match_blocking_path(native_lib):
    """Several async contexts sharing the per-device warp stream, driven through hf_interpolate_period with
    zero-copy device frames, produce the same frames as the blocking reference-style call sequence."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, OpticalFlowCalcSDR
    H, W, n = 180, 320, 6
    frames = [synth.Scene(H, W, False, 300 + s) for s in range(3)]
    frames = [[sc.frame(k) for k in range(n)] for sc in frames]
    ts = [0.0, 0.3996, 0.7992]
    # blocking ground truth
    want = []
    for fs in frames:
        c = OpticalFlowCalcSDR(H, W, search_radius=12)
        outs = []
        for k, f in enumerate(fs):
            c.updateFrame(f)
            if k >= 2:
                c.calculateOpticalFlow()
                for t in ts:
                    c.warpFrames(t, 2)
                    outs.append(c.downloadFrame().copy())
        want.append(outs)
        c.close()
    A = capi.HF_FLAG_ASYNC
    for extra in (0, capi.HF_FLAG_DUAL_STREAM, capi.HF_FLAG_NO_FUSED_WARP, capi.HF_FLAG_NO_TIMING,
                  capi.HF_FLAG_NO_GRAPH, capi.HF_FLAG_NO_LAZY_ARGMIN, capi.HF_FLAG_PROFILE,
                  capi.HF_FLAG_DUAL_STREAM | capi.HF_FLAG_NO_TIMING | capi.HF_FLAG_PROFILE,
                  capi.HF_FLAG_NO_GRAPH | capi.HF_FLAG_NO_LAZY_ARGMIN | capi.HF_FLAG_NO_FUSED_WARP):
        run_batched(frames, want, ts, A | extra)


def run_batched(frames, want, ts, flags):
    from hopperrender_amd.calc import DeviceBuffer, OpticalFlowCalcSDR
    H, W, n = 180, 320, len(frames[0])
    calcs = [OpticalFlowCalcSDR(H, W, search_radius=12, flags=flags) for _ in frames]
    dev = [[DeviceBuffer(f.nbytes) for f in fs] for fs in frames]
    for s, fs in enumerate(frames):
        for k, f in enumerate(fs):
            dev[s][k].upload(f)
    outbufs = [[[DeviceBuffer(calcs[0].output_frame_bytes) for _ in ts] for _ in range(n)] for _ in frames]
    for k in range(n):
        for s, c in enumerate(calcs):
            if k < 2:
                c.updateFrameDeviceRef(dev[s][k].ptr)
            else:
                c.interpolatePeriod(dev[s][k].ptr, ts, [b.ptr for b in outbufs[s][k]], 2)
    for c in calcs:
        c.sync()
    for s in range(len(frames)):
        got = [outbufs[s][k][i].download(np.uint8) for k in range(2, n) for i in range(len(ts))]
        assert len(got) == len(want[s])
        for a, b in zip(got, want[s]):
            assert (a == b).all()
    for c in calcs:
        c.close()


def test_rs4_4320p_matches_oracle(native_lib):
    """7680x4320 (resolution scalar 4, the largest grid the reference's auto mode produces is still 480x270)."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    from oracle import oracle
    H, W, R = 4320, 7680, 16
    sc = synth.Scene(H, W, False, 404)
    f = [sc.frame(k) for k in range(3)]
    g = oracle.make_geom(0, H, W)
    assert (g.rs, g.lw, g.lh) == (4, 480, 270)
    off_o, blur_o, tot_o, oob = oracle.calculate_optical_flow(f[1], f[2], g, R)
    c = OpticalFlowCalcSDR(H, W, search_radius=R)
    for x in f:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    assert oob == 0
    assert (c.readOffsets() == off_o).all() and (c.readBlurredFlow(1) == blur_o).all() and c.m_totalFrameDelta == tot_o
    c.calculateOpticalFlow()
    c.warpFrames(0.5994, 2)
    assert (c.downloadFrame() == oracle.warp_frames(f[0], f[1], blur_o, g, 0.5994, 2)).all()
    c.close()


def test_calc_time_statistics_contract(native_lib):
    """m_ofcCalcTime / Avg / Peak bookkeeping of opticalFlowCalcSDR.cpp:125-138 (CALC_TIME_INTERVAL = 240)."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    sc = synth.Scene(36, 64, False, 1)
    c = OpticalFlowCalcSDR(36, 64)
    for k in range(3):
        c.updateFrame(sc.frame(k))
    for i in range(240):
        c.calculateOpticalFlow()
    assert c.m_ofcCalcTime > 0 and c.m_ofcPeakCalcTime >= c.m_ofcCalcTime
    assert c.m_ofcAvgCalcTime == 0.0           # the average is only published after 240 calls
    c.calculateOpticalFlow()                   # 241st call closes the first interval
    assert c.m_ofcAvgCalcTime > 0 and c.m_ofcPeakCalcTime == c.m_ofcCalcTime
    c.warpFrames(0.5, 2)
    c.downloadFrame()
    assert c.m_warpCalcTime > 0                # first warp launch -> end of the readback (:36-41)
    c.close()


def test_recreate_on_format_change_and_threads(native_lib):
    """The filter deletes and lazily recreates the calculator on every resolution/stride change
    (HopperRender.cpp:762-765,856-859): construction/destruction must be cheap, leak-free and contexts must be
    independent across host threads."""
    import threading
    import time
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    sizes = [(0, 180, 320), (1, 360, 640), (0, 338, 600), (1, 180, 320)]
    frames = [synth.random_frame(H, W, bool(hdr), seed=i) for i, (hdr, H, W) in enumerate(sizes)]   # not part of the timing
    t0 = time.perf_counter()
    for i in range(24):
        hdr, H, W = sizes[i % len(sizes)]
        c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W)
        sc = frames[i % len(sizes)]
        for _ in range(3):
            c.updateFrame(sc)
        c.calculateOpticalFlow()
        c.warpFrames(0.5, 2)
        c.downloadFrame()
        c.close()
    assert (time.perf_counter() - t0) / 24 < 1.0     # reference: hundreds of ms of OpenCL JIT per construction; here ~10 ms
                                                     # (generous bound: a cold box pages the library in during the first iterations)

    results, errors = {}, []

    def worker(idx, hdr, H, W, seed):
        try:
            sc = synth.Scene(H, W, bool(hdr), seed)
            f = [sc.frame(k) for k in range(3)]
            c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=10)
            for _ in range(5):
                c.m_frameCount = 0
                for x in f:
                    c.updateFrame(x)
                c.calculateOpticalFlow()
            results[idx] = (c.readOffsets(), c.m_totalFrameDelta, f, hdr, H, W)
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i, i % 2, 180, 320, 50 + i)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for idx, (off, tot, f, hdr, H, W) in results.items():
        o, _, t_o, _ = oracle.calculate_optical_flow(f[1], f[2], oracle.make_geom(hdr, H, W), 10)
        assert (off == o).all() and tot == t_o


def test_async_host_io_pipeline_matches_blocking(native_lib):
    """hf_update_frame_async / hf_download_frame_async (pinned buffers, side streams, output ring) against the
    blocking reference-style sequence, single- and dual-stream contexts."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, PinnedArray
    H, W, n = 180, 320, 7
    sc = synth.Scene(H, W, True, 77)
    frames = [sc.frame(k) for k in range(n)]
    ts = [0.0, 0.1998, 0.3996, 0.5994, 0.7992]
    c = OpticalFlowCalcHDR(H, W, search_radius=9)
    want = []
    for k, f in enumerate(frames):
        c.updateFrame(f)
        if k >= 2:
            c.calculateOpticalFlow()
            for t in ts:
                c.warpFrames(t, 2)
                want.append(c.downloadFrame().copy())
    c.close()
    for flags in (capi.HF_FLAG_ASYNC, capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM):
        c = OpticalFlowCalcHDR(H, W, search_radius=9, flags=flags)
        pin_in = [PinnedArray(f.size, np.uint16) for f in frames]
        for p, f in zip(pin_in, frames):
            p.array[:] = f
        n_el = c.output_frame_bytes // 2
        pin_out = [PinnedArray(n_el, np.uint16) for _ in range(len(want))]
        o = 0
        for k in range(n):
            c.updateFrameAsync(pin_in[k])
            if k >= 2:
                c.calculateOpticalFlow()
                for t in ts:
                    c.warpFrames(t, 2)
                    c.downloadFrameAsync(pin_out[o])
                    o += 1
        c.sync()
        for i, w in enumerate(want):
            assert (pin_out[i].array == w).all(), f"flags {flags} output {i}"
        c.close()
        for p in pin_in + pin_out:
            p.free()


@pytest.mark.parametrize("hdr,H,W,max_res", [(0, 568, 1388, 1000), (1, 300, 1100, 600)])
def test_wide_grid_large_windows_with_neighbour_term(native_lib, hdr, H, W, max_res):
    """Grids wider than 512 (maxCalcRes above the default 270): the levels that carry the neighbour bias
    (iteration >= 4, calcDeltaSumsKernelSDR.h:3,112) still have windows > 32, which take the partial-sum + explicit
    argmin path.  Every iteration count, so that each level is the last one once."""
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    sc = synth.Scene(H, W, bool(hdr), seed=4242)
    f = [sc.frame(k) for k in range(3)]
    g = oracle.make_geom(hdr, H, W, 0, 0, max_res)
    assert g.lw > 512
    for it in (4, 5, 6, 7, 0):
        c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, 0, 0, 3, 4, 0.0, 255.0, max_res, iterations=it, search_radius=12)
        for x in f:
            c.updateFrame(x)
        c.calculateOpticalFlow()
        off_o, blur_o, tot_o, oob = oracle.calculate_optical_flow(f[1], f[2], g, 12, it, 3, 4, 4)
        assert oob == 0
        assert (c.readOffsets() == off_o).all(), it
        assert (c.readBlurredFlow(1) == blur_o).all(), it
        assert c.m_totalFrameDelta == tot_o
        c.close()


def test_threads_capture_and_readback_stress(native_lib):
    """Several host threads creating contexts, capturing their flow graphs and reading results back at the same time.
    The library must stay off the legacy (null) stream: a synchronous hipMemcpy there while another thread captures
    invalidates that capture (HIP error 906).  tools/stress_threads.py, 6 rounds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_threads.py"), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]


@pytest.mark.parametrize("hdr,H,W", [(0, 1080, 1920), (1, 2160, 3840)])
def test_dual_stream_first_period_with_copied_in_frames(native_lib, hdr, H, W):
    """HF_FLAG_DUAL_STREAM: the very first period's warps (the filter warps as soon as m_frameCount >= 3, i.e. right after
    the first flow calculation, when no flow buffer carries a tag yet) must still be ordered behind the uploads of the
    frames they read.  Frames are COPIED into the ring (hf_update_frame_device), large ones, so a missing dependency
    would read a half-written ring slot."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    sc = synth.Scene(H, W, bool(hdr), 11)
    frames = [sc.frame(k) for k in range(4)]
    dev = [DeviceBuffer(f.nbytes) for f in frames]
    for d, f in zip(dev, frames):
        d.upload(f)
    ts = [0.0, 0.3996, 0.7992]
    dt = np.uint16 if hdr else np.uint8
    results = []
    for flags in (0, capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM):
        for rep in range(3 if flags else 1):
            c = cls(H, W, search_radius=8, flags=flags)
            outs = [DeviceBuffer(c.output_frame_bytes) for _ in range(2 * len(ts))]
            for k in range(3):
                c.updateFrameDevice(dev[k].ptr)
            c.calculateOpticalFlow()
            c.interpolateOnly(ts, [o.ptr for o in outs[:3]], 2)         # first period: previous flow = zero, frames 0 / 1
            c.updateFrameDevice(dev[3].ptr)                              # overwrites the oldest ring slot behind those warps
            c.calculateOpticalFlow()
            c.interpolateOnly(ts, [o.ptr for o in outs[3:]], 2)
            c.sync()
            results.append([o.download(dt) for o in outs])
            c.close()
    for r in results[1:]:
        for a, b in zip(results[0], r):
            assert (a == b).all()
