"""GPU: the device-side diagnostic counters (include/hopperflow_diag.h hf_debug_counters_*): what the chain's table kernels and the staged
period warp decide per window / per workgroup.  The reuse counts must equal the CPU model of that decision on the oracle's chain
(tests/flow_reuse_model.py) window for window; results must not change while the counters are on."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["bench", "static", "chaotic"])
def test_reuse_counters_match_the_cpu_model(native_lib, scene):
    from flow_reuse_model import reuse_shares
    from hopperrender_amd import synth
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    from oracle import oracle
    H, W = 1080, 1920
    sc = synth.ContentScene(scene, H, W, False, 1234)
    f = [sc.frame(i) for i in range(3)]
    g = oracle.make_geom(0, H, W)
    from hopperrender_amd import capi
    c = OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_SAD_REUSE_ALWAYS)   # (chaotic content would switch the tables off)
    for x in f:
        c.updateFrame(x)
    c.calculateOpticalFlow(); c.sync()
    before = c.readOffsets()
    c.countersEnable(True)
    c.calculateOpticalFlow(); c.sync()          # same ring: the same pair again
    cc = c.counters(reset=True)
    assert (c.readOffsets() == before).all()
    model = {(ws, ax): s for ws, ax, s in reuse_shares(f[1], f[2], g)}
    full_tiles = (g.lw // 32) * (g.lh // 32)
    assert set(cc["levels"]) == {32, 16, 8, 4, 2}
    for ws, lv in cc["levels"].items():
        for ai, ax in enumerate("XY"):
            windows, reused = lv[ax]
            assert windows == full_tiles * (32 // ws) ** 2, (ws, ax, windows)
            assert abs(reused / windows - model[(ws, ai)]) < 1e-9, (scene, ws, ax, reused / windows, model[(ws, ai)])
    assert c.counters()["levels"] == {}            # reset
    c.countersEnable(False)
    c.calculateOpticalFlow(); c.sync()
    assert (c.readOffsets() == before).all()
    c.close()


def test_warp_workgroup_counters_of_a_batch(native_lib):
    """Four 2160p HDR members through hf_batch_run_period: the fused period warp's workgroups are counted in the leader, by path."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    H, W, n = 2160, 3840, 4
    sc = synth.ContentScene("bench", H, W, True, 77)
    frames = [sc.frame(i) for i in range(5)]
    dev = [DeviceBuffer(f.nbytes) for f in frames]
    for d, f in zip(dev, frames):
        d.upload(f)
    cs = [OpticalFlowCalcHDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    outs = [[DeviceBuffer(c.output_frame_bytes) for _ in range(5)] for c in cs]
    b = FlowBatch(cs)
    ts = [[0.1, 0.3, 0.5, 0.7, 0.9]] * n
    optr = [[o.ptr for o in row] for row in outs]
    for i in range(4):
        b.runPeriod(b.preparePeriod([dev[i].ptr] * n, ts if i >= 2 else None, optr, 2, calculate_flow=i >= 2))
    b.sync()
    ref = outs[1][2].download(np.uint16)
    cs[0].countersEnable(True)
    b.runPeriod(b.preparePeriod([dev[4].ptr] * n, ts, optr, 2))
    b.sync()
    cc = cs[0].counters()
    wg = cc["warp_workgroups"]
    if sum(wg.values()):       # the staged kernel ran (it does for batches of frames above 1080p)
        assert wg["staged"] / sum(wg.values()) > 0.8, wg
    assert cc["levels"] and all(v["X"][0] % n == 0 for v in cc["levels"].values())     # every member counted in the leader
    cs[0].countersEnable(False)
    b.close()
    assert ref.size == outs[1][2].download(np.uint16).size
    for c in cs:
        c.close()


def test_tables_switch_off_on_hostile_content_and_back(native_lib):
    """The chain reports how many 32-windows kept their offsets; while hardly any does, the following chains of a batch run without SAD tables
    (hf_stats sad_tables / still_share), with identical results, and come back when the content calms down.  (Fewer than four pairs per
    launch never keep tables: a launch of a nearly idle device is as long as its slowest wave.)"""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR
    from oracle import oracle
    H, W, n = 1080, 1920, 4
    g = oracle.make_geom(0, H, W)
    calm = synth.ContentScene("bench", H, W, False, 21)
    wild = synth.ContentScene("chaotic", H, W, False, 22)
    seq = [calm.frame(i) for i in range(4)] + [wild.frame(i) for i in range(6)] + [calm.frame(i) for i in range(4, 10)]
    lone = OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16)
    cs = [OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    b = FlowBatch(cs)
    modes = []
    for i, fr in enumerate(seq):
        for c in cs + [lone]:
            c.updateFrame(fr)
        if i < 2:
            continue
        b.calculateOpticalFlow(); b.sync()
        lone.calculateOpticalFlow(); lone.sync()
        modes.append(cs[0].stats()["sad_tables"])
        assert lone.stats()["sad_tables"] == 0
        off, _, tot, _ = oracle.calculate_optical_flow(seq[i - 1], seq[i], g, 16, 0, 8, 6, 4)
        for c in cs + [lone]:
            assert (c.readOffsets() == off).all() and c.m_totalFrameDelta == tot, i
    assert modes[0] == 1 and modes[1] == 1                   # calm start
    assert 0 in modes[3:8], modes                            # hostile stretch: tables off after the reports arrive
    assert modes[-1] == 1, modes                             # and on again
    b.close()
    for c in cs + [lone]:
        c.close()
