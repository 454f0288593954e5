"""GPU: hf_batch_timeline_* -- the start / stop of every dispatch of a batch's periods on the device clock, without a profiler
(what profiles/r05_pipeline_timeline.json is made from).  While it records, the chain is issued launch by launch instead of as a
graph replay: the results must not change."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_timeline_records_every_dispatch_and_changes_nothing(native_lib):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    H, W, n, R = 2160, 3840, 4, 16
    sc = synth.Scene(H, W, True, 99)
    frames = [sc.frame(k) for k in range(6)]
    dev = []
    for f in frames:
        b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    T = [0.1988, 0.3996, 0.5984, 0.7992, 0.998]

    def run(timeline):
        ms = [OpticalFlowCalcHDR(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for _ in range(n)]
        b = FlowBatch(ms)
        outs = [[DeviceBuffer(ms[0].output_frame_bytes) for _ in range(5)] for _ in range(n)]
        if timeline:
            b.timelineEnable(3 * 14 + 20, 2)          # periods 0 and 1 pass unobserved, then three periods (14 dispatches each) are recorded
        for k in range(6):
            b.runPeriod(b.preparePeriod([dev[k].ptr] * n, [T] * n if k >= 2 else None, [[x.ptr for x in o] for o in outs] if k >= 2 else None, 2,
                                        calculate_flow=k >= 1))
        b.sync()
        recs = b.timelineRead() if timeline else None
        flows = [m.readBlurredFlow(1) for m in ms]
        frames_out = [o[3].download(np.uint16) for o in outs]
        b.close()
        for m in ms:
            m.close()
        for o in outs:
            for x in o:
                x.free()
        return recs, flows, frames_out

    recs, flows_t, out_t = run(True)
    _, flows_p, out_p = run(False)
    for a, b in zip(flows_t + out_t, flows_p + out_p):
        assert np.array_equal(a, b)
    periods = sorted({r[1] for r in recs})
    assert periods == [0, 1, 2]                        # then it ran out of whole periods' worth of records and went back to graph replays
    names = [r[0] for r in recs if r[1] == 1]
    assert names[0] == "grid_samples" and names[1] == "warp_period"        # deferred planes: the warp goes out ahead of the period's chain
    chain = names[2:]
    assert chain == ["large_windows_x", "large_windows_y"] * 3 + ["level_32", "level_16", "level_8", "level_4", "level_2", "blur"], chain
    t = [(r[2], r[3]) for r in recs]
    assert all(e > s for s, e in t)
    assert all(t[i + 1][0] >= t[i][1] - 1e-3 for i in range(len(t) - 1))   # one stream: dispatches do not overlap
    warp = [e - s for (k, p, s, e) in recs if k == "warp_period"]
    assert all(0.05 < w < 5.0 for w in warp), warp                         # milliseconds: four members' fused periods


def test_timeline_off_and_argument_checks(native_lib):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR
    ms = [OpticalFlowCalcSDR(180, 320, flags=capi.HF_FLAG_ASYNC) for _ in range(2)]
    b = FlowBatch(ms)
    assert b.timelineRead() == []
    with pytest.raises(capi.HopperFlowError):
        b.timelineEnable(-1)
    b.timelineEnable(64); b.timelineEnable(0)
    assert b.timelineRead() == []
    b.close()
    for m in ms:
        m.close()


def test_clock_probe_reads_a_plausible_shader_clock(native_lib):
    import ctypes as C
    mhz = C.c_double(0.0)
    assert native_lib.hf_clock_probe(0, 200, C.byref(mhz)) == 0
    assert 100.0 < mhz.value < 2600.0, mhz.value          # idle: a low DPM level; under load: up to the 2.4 GHz peak
    assert native_lib.hf_clock_probe(0, 5, C.byref(mhz)) != 0


def test_timeline_of_an_eager_plane_batch(native_lib):
    """1080p SDR batches build their planes eagerly: a period is the plane kernel, the chain's twelve launches, then the period warp."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcSDR
    H, W, n = 1080, 1920, 6
    sc = synth.Scene(H, W, False, 7)
    dev = []
    for k in range(5):
        f = sc.frame(k); b = DeviceBuffer(f.nbytes); b.upload(f); dev.append(b)
    ms = [OpticalFlowCalcSDR(H, W, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for _ in range(n)]
    b = FlowBatch(ms)
    assert not b.defersPlanes()
    outs = [[DeviceBuffer(ms[0].output_frame_bytes) for _ in range(3)] for _ in range(n)]
    T = [0.1988, 0.5984, 0.998]
    b.timelineEnable(2 * 14 + 20, 2)
    for k in range(5):
        b.runPeriod(b.preparePeriod([dev[k].ptr] * n, [T] * n if k >= 2 else None, [[x.ptr for x in o] for o in outs] if k >= 2 else None, 2, calculate_flow=k >= 1))
    b.sync()
    recs = b.timelineRead()
    names = [r[0] for r in recs if r[1] == 0]
    assert names == ["plane"] + ["large_windows_x", "large_windows_y"] * 3 + ["level_32", "level_16", "level_8", "level_4", "level_2", "blur", "warp_period"], names
    b.close()
    for m in ms:
        m.close()
    for x in dev + [y for o in outs for y in o]:
        x.free()


def test_hbm_copy_probe_reads_a_plausible_rate(native_lib):
    import ctypes as C
    g = C.c_double(0.0)
    assert native_lib.hf_hbm_copy_probe(0, 1 << 30, 3, C.byref(g)) == 0
    assert 2000.0 < g.value < 8000.0, g.value           # GB/s read + write: HBM3E at 8 TB/s peak
    assert native_lib.hf_hbm_copy_probe(0, 16, 3, C.byref(g)) != 0
