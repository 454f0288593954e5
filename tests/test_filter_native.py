"""CPU: the native caller-protocol code (hopperrender_amd/csrc/hf_filter.cpp, hf_filter_* C ABI) against the independent
pure-Python restatement of the same reference lines (hopperrender_amd/protocol.py): blend schedule
(HopperRender.cpp:944-948,1192-1197), scene-change detector (:959-972,1126-1176), governor (:1438-1463)."""
import numpy as np
import pytest

from hopperrender_amd.protocol import (MAX_SEARCH_RADIUS, MIN_SEARCH_RADIUS, SOURCE_24, TARGET_60, TARGET_120, BlendSchedule,
                                       NativeFilter, SceneChangeDetector, auto_adjust_radius)


@pytest.mark.parametrize("target", [TARGET_60, TARGET_120, 100000, 400000])
def test_native_schedule_matches_python(native_lib, target):
    nf = NativeFilter(SOURCE_24, target)
    py = BlendSchedule(SOURCE_24, target, active=SOURCE_24 > target)   # UpdateInterpolationStatus: only when the target rate is higher
    for _ in range(500):
        n = nf.begin_source_frame()
        assert n == py.begin_source_frame()
        for _ in range(n):
            assert nf.next_scalar() == py.next_scalar()            # same doubles, same order of operations


def test_native_schedule_follows_the_playback_rate(native_lib):
    nf = NativeFilter(SOURCE_24, TARGET_60)
    nf.new_segment(2.0)                                            # NewSegment: playback frame time = source / rate (:836)
    py = BlendSchedule(int(SOURCE_24 * 0.5), TARGET_60)
    assert nf.state()["playback_frame_time"] == int(SOURCE_24 * (1.0 / 2.0))
    for _ in range(50):
        n = nf.begin_source_frame()
        assert n == py.begin_source_frame()
        for _ in range(n):
            assert nf.next_scalar() == py.next_scalar()
    nf.new_segment(4.0)                                            # 96 fps source, 60 fps target: nothing to interpolate
    assert nf.state()["active"] == 0 and nf.begin_source_frame() == 1


@pytest.mark.parametrize("seed", range(12))
def test_native_scene_change_matches_python(native_lib, seed):
    rng = np.random.default_rng(seed)
    thr = int(rng.choice([50, 200, 1000]))
    nf = NativeFilter(SOURCE_24, TARGET_60, scene_change_threshold=thr)
    py = SceneChangeDetector(SOURCE_24, thr)
    frame = 3
    hits = 0
    for step in range(400):
        base = int(rng.integers(0, 400))
        delta = base + (int(rng.integers(300, 5000)) if rng.random() < 0.08 else 0)
        if rng.random() < 0.02:                                    # a seek: histories start over, frame counter restarts
            nf.new_segment(1.0); py.reset(); frame = 3
        nf.push(frame, delta); py.push(frame, delta)
        for _ in range(int(rng.integers(1, 4))):                   # evaluated once per output frame (:1126)
            a, b = nf.detect(frame), py.detect(frame)
            assert a == b, (step, frame)
            hits += a
        st = nf.state()
        assert (st["peak_scene_change_delta"], st["peak_scene_change_delta2"]) == (py.peak_delta, py.peak_delta2)
        assert st["frame_delta_history"] == len(py.frame_delta_history)
        frame += 1
    assert hits > 0 or thr == 1000


def test_native_scene_change_known_answer(native_lib):
    nf = NativeFilter(SOURCE_24, TARGET_60, scene_change_threshold=200)
    hits = []
    for i, delta in enumerate([100, 110, 105, 900, 120, 100]):
        nf.push(i + 3, delta)
        hits.append(nf.detect(i + 3))
    assert hits == [False, False, False, False, True, False]      # the spike is "current" when it sits second-to-last
    st = nf.state()
    # spike as "current": average of the <= 10 entries up to and including it = (900 + 105 + 110) // 3 = 371 (:1136-1140)
    assert st["peak_scene_change_delta"] == 900 - 371 and st["peak_scene_change_delta2"] == 900 - 120
    assert st["scene_change_delta1"] == 120 - (120 + 900 + 105 + 110) // 4 and st["scene_change_delta2"] == 120 - 100


def test_native_governor_matches_python(native_lib):
    rng = np.random.default_rng(7)
    nf = NativeFilter(SOURCE_24, TARGET_60)
    r = MIN_SEARCH_RADIUS
    for _ in range(300):
        ofc, warp = float(rng.uniform(0, 0.04)), float(rng.uniform(0, 0.01))
        nf.add_warp_duration(warp * 0.25); nf.add_warp_duration(warp * 0.75)
        want = auto_adjust_radius(ofc, warp * 0.25 + warp * 0.75, SOURCE_24, r)
        r = nf.auto_adjust(ofc, r)
        assert r == want and MIN_SEARCH_RADIUS <= r <= MAX_SEARCH_RADIUS
        assert nf.state()["total_warp_duration"] == 0.0            # reset for the next source frame (:1462)
    # plenty of headroom walks 5 -> 16 one step per source frame and stays; an impossible period walks back to 5
    r = MIN_SEARCH_RADIUS
    seq = []
    for _ in range(14):
        r = nf.auto_adjust(0.0003, r); seq.append(r)
    assert seq == [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 16, 16]
    nf.set_playback_frame_time(1000)
    seq = []
    for _ in range(13):
        r = nf.auto_adjust(0.0003, r); seq.append(r)
    assert seq == [15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 5, 5]
