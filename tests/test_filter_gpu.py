"""GPU, end to end: the native caller protocol (hf_filter_*, hopperrender_amd/csrc/hf_filter.cpp) around the real calculator.
  * a clip with a hard cut through tests/cpp/replay_filter.cpp: the period whose two source frames straddle the cut is
    copyFrame output (bit-equal to the oracle's copy), its neighbours are warps (HopperRender.cpp:1126-1183);
  * the governor on the calculator's own m_ofcCalcTime / m_warpCalcTime: 5 -> 16 on an MI355X, and back down when the
    source period is set artificially short (:1438-1463);
  * ownership of the public fields of the C++ adapter (a settings-thread write during a blocking call survives);
  * hf_filter_deliver (the whole DeliverToRenderer inside the library) against the step-by-step replay."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_exe(tmp_path, name):
    from hopperrender_amd import build
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", build.LIBDIR, "-lopticalflowcalc", "-lhopperflow", f"-Wl,-rpath,{build.LIBDIR}"])
    return exe


def cut_clip(H, W, hdr, n_before, n_after, seed=42):
    from hopperrender_amd import synth
    a = synth.Scene(H, W, bool(hdr), seed)
    b = synth.Scene(H, W, bool(hdr), seed + 999)
    return [a.frame(k) for k in range(n_before)] + [b.frame(n_before + k) for k in range(n_after)]


@pytest.mark.parametrize("hdr,target", [(0, 166667), (1, 83333)])
def test_scene_cut_turns_the_cut_period_into_copies(native_lib, tmp_path, hdr, target):
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule, SceneChangeDetector
    from oracle import oracle
    exe = build_exe(tmp_path, "replay_filter")
    H, W, R = 180, 320, 8
    frames = cut_clip(H, W, hdr, 7, 5)          # frames 0..6 scene A, 7..11 scene B: the pair (6, 7) is the cut
    n = len(frames)
    g = oracle.make_geom(hdr, H, W)
    flows = {k: oracle.calculate_optical_flow(frames[k - 1], frames[k], g, R) for k in range(2, n)}   # keyed by the newer frame
    deltas = {k: flows[k][2] for k in flows}
    # the divisor of m_totalFrameDelta is 10 for SDR and 6 for HDR (opticalFlowCalcSDR.cpp:93 / opticalFlowCalcHDR.cpp:93):
    # pick the threshold from the clip itself so the test exercises the decision, not the content generator
    # (the synthetic scenes move fast -- the no-motion candidate the delta is taken from already scores ~3000 -- so the
    # cut is a spike of a few hundred on top, like a real cut between two busy scenes)
    spike, around = deltas[7], max(v for k, v in deltas.items() if k != 7)
    assert spike > around + 100, deltas
    hist = [deltas[k] for k in range(2, 8)]                                   # what the detector averages when the spike is "current"
    d1, d2 = spike - sum(hist[::-1][:10]) // len(hist), spike - deltas[8]     # HopperRender.cpp:1136-1144
    threshold = max(min(d1, d2) // 2, 1)
    for k, f in enumerate(frames):
        f.tofile(str(tmp_path / f"in{k}.bin"))
    r = subprocess.run([exe, str(hdr), str(H), str(W), str(n), str(tmp_path / "in"), str(tmp_path / "out"), str(target), str(R), str(threshold)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in r.stdout.splitlines()]
    src = [l for l in lines if l and l[0] == "src"]
    outl = [l for l in lines if l and l[0] == "out"]
    assert [int(l[5]) for l in src[2:]] == [deltas[k] for k in range(2, n)]      # m_totalFrameDelta per source frame
    # independent prediction: the Python restatement of the detector on the oracle's deltas
    det = SceneChangeDetector(SOURCE_24, threshold)
    plan = BlendSchedule(SOURCE_24, target).plan(n)
    dt = np.uint16 if hdr else np.uint8
    out_index, copies_after_warmup = 0, []
    for k in range(n):
        count = k + 1
        if count >= 3:
            det.push(count, deltas[k])
        for t in plan[k]:
            cut = det.detect(count)
            got = np.fromfile(str(tmp_path / f"out{out_index}.bin"), dtype=dt)
            if count >= 3 and not cut:
                prev = flows[k - 1][1] if (k - 1) in flows else np.zeros((2, g.lh, g.lw), np.int16)
                ref = oracle.warp_frames(frames[k - 2], frames[k - 1], prev, g, np.float32(t), 2)
                assert outl[out_index][2] == "warp", (k, outl[out_index])
            else:
                idx_frame = frames[k - 2] if count >= 3 else (frames[k - 1] if count == 2 else frames[k])   # opticalFlowCalcSDR.cpp:173
                ref = oracle.copy_frame(idx_frame, g)
                assert outl[out_index][2] == "copy", (k, outl[out_index])
                if count >= 3:
                    copies_after_warmup.append(k)
                    assert outl[out_index][3] == "1"
            assert (got == ref).all(), f"output {out_index} (source frame {k}, t={t})"
            out_index += 1
    # the spike is "current" (second to last in the history) when source frame 8 arrives: that period shows frames
    # 6 -> 7 = the cut, and only that period is copied
    assert 8 in copies_after_warmup and set(copies_after_warmup) <= {8, 9}, copies_after_warmup
    assert len(outl) == out_index


def test_governor_walks_the_radius_on_real_timings(native_lib, tmp_path):
    """autoAdjustSettings on the calculator's own timings (HopperRender.cpp:1438-1463): with a 41.7 ms source period the
    MI355X has headroom at every radius, so R climbs 5 -> 16 one step per source frame and stays; with a source period no
    GPU can meet (100 us for a blocking upload + flow calculation + 2.5 warps and readbacks) it walks back down to 5."""
    from hopperrender_amd import synth
    exe = build_exe(tmp_path, "replay_filter")
    H, W, n = 360, 640, 16
    sc = synth.Scene(H, W, False, 5)
    for k in range(n):
        sc.frame(k % 6).tofile(str(tmp_path / f"in{k}.bin"))

    def run(target, radius, short_after, playback):
        r = subprocess.run([exe, "0", str(H), str(W), str(n), str(tmp_path / "in"), str(tmp_path / "out"), str(target), str(radius), "200", "1",
                            str(short_after), str(playback)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        src = [l.split() for l in r.stdout.splitlines() if l.startswith("src")]
        return [int(l[3]) for l in src], [float(l[7]) for l in src[3:]]

    radii, ofc_us = run(166667, 0, -1, 0)                         # 23.976 -> 60 fps, radius starts at MIN_SEARCH_RADIUS
    assert radii == [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 16, 16, 16, 16], radii
    assert all(0.0 < t < 5000.0 for t in ofc_us), ofc_us          # real device timings, far below the 41.7 ms period
    radii, ofc_us = run(400, 16, 0, 1000)                         # a 100 us source period, 2.5 outputs each, radius starts at 16
    assert radii == [16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 5, 5, 5, 5], radii   # (nothing measured yet at the first frame)
    assert all(t > 0.0 for t in ofc_us)


def test_settings_written_during_a_blocking_call_survive(native_lib, tmp_path):
    from hopperrender_amd import synth
    from oracle import oracle
    exe = build_exe(tmp_path, "field_ownership")
    H, W = 360, 640
    sc = synth.Scene(H, W, False, 77)
    frames = [sc.frame(k) for k in range(3)]
    for k, f in enumerate(frames):
        f.tofile(str(tmp_path / f"in{k}.bin"))
    r = subprocess.run([exe, str(H), str(W), str(tmp_path / "in"), str(tmp_path / "flow.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kv = dict(l.split() for l in r.stdout.splitlines() if len(l.split()) == 2)
    assert int(kv["calls"]) > 200                                  # the write landed while calls were in flight
    assert (kv["m_deltaScalar"], kv["m_opticalFlowSearchRadius"], kv["m_outputWhiteLevel"]) == ("5", "9", "200.0")
    g = oracle.make_geom(0, H, W)
    _, blur, tot, oob = oracle.calculate_optical_flow(frames[1], frames[2], g, 9, 0, 5, 6, 4)
    assert oob == 0
    flow = np.fromfile(str(tmp_path / "flow.bin"), dtype=np.int16).reshape(2, g.lh, g.lw)
    assert (flow == blur).all() and int(kv["m_totalFrameDelta"]) == tot    # the NEXT call used delta scalar 5 and radius 9
    assert kv["m_frameCount_before"] == "3" and kv["m_frameCount_after_new_segment"] == "1"


@pytest.mark.parametrize("hdr", [0, 1])
def test_deliver_in_the_library_equals_the_step_by_step_replay(native_lib, hdr):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, TARGET_120, FilterReplay, NativeFilter
    H, W = 180, 320
    frames = cut_clip(H, W, hdr, 6, 4, seed=7)
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    a, b = cls(H, W, search_radius=8), cls(H, W, search_radius=8)
    step = FilterReplay(a, SOURCE_24, TARGET_120, scene_change_threshold=20)
    whole = NativeFilter(SOURCE_24, TARGET_120, scene_change_threshold=20)
    kinds_all = []
    for f in frames:
        want = step.deliver(f)
        got, kinds = whole.deliver(b, f)
        kinds_all += kinds
        assert len(got) == len(want)
        for x, y in zip(got, want):
            assert (x == y).all()
    assert kinds_all == [k for k, _ in step.log]
    assert "copy" in kinds_all[12:] and "warp" in kinds_all     # the cut produced copies after the warm-up frames
    a.close(); b.close()


@pytest.mark.parametrize("cut_at", [20, 27, 14])
def test_sharded_timeline_equals_the_sequential_run(native_lib, cut_at):
    """hopperrender_amd/batch.py: a clip cut into contiguous chunks for 3 ranks reproduces the sequential filter output
    for output -- including the scene-change decision, whose delta history a chunk rebuilds by replaying the flow of
    the 12 periods before its first one (no exchange between ranks).  cut_at 27 / 14: the cut sits right at a chunk
    border (chunks start at periods 0, 14, 27), so the decisive deltas belong to the preceding rank's periods."""
    from hopperrender_amd import batch
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, TARGET_60, FilterReplay
    H, W, n, world, thr = 180, 320, 40, 3, 150
    frames = cut_clip(H, W, 0, cut_at, n - cut_at, seed=5)
    seq = OpticalFlowCalcSDR(H, W, search_radius=8)
    replay = FilterReplay(seq, SOURCE_24, TARGET_60, scene_change_threshold=thr)
    want = []
    for f in frames:
        want += [o.copy() for o in replay.deliver(f)]
    want_kinds = [k for k, _ in replay.log]
    assert "copy" in want_kinds[6:], "the clip must contain a detected scene change"
    got, got_kinds = [], []
    for rank in range(world):
        chunk = batch.shard_timeline(n, world, rank, SOURCE_24, TARGET_60)
        c = OpticalFlowCalcSDR(H, W, search_radius=8)
        outs, kinds = batch.run_chunk(c, chunk, frames, scene_change_threshold=thr)
        assert chunk.first_output == len(got)
        got += outs; got_kinds += kinds
        c.close()
    assert got_kinds == want_kinds
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert (a == b).all(), i
    seq.close()
