"""The C++ drop-in surface (include/opticalFlowCalc.h) driven exactly like the reference's filter drives
the reference class (tests/cpp/replay_filter.cpp), checked frame by frame against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hdr,target", [(0, 166667), (1, 83333)])
def test_replay_filter_matches_oracle(native_lib, tmp_path, hdr, target):
    from hopperrender_amd import build, synth
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule
    from oracle import oracle
    exe = str(tmp_path / "replay_filter")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "replay_filter.cpp"), "-o", exe,
                           "-L", build.LIBDIR, "-lopticalflowcalc", "-lhopperflow",
                           f"-Wl,-rpath,{build.LIBDIR}"])
    H, W, n, R = 180, 320, 6, 9
    sc = synth.Scene(H, W, bool(hdr), seed=99)
    frames = [sc.frame(k) for k in range(n)]
    for k, f in enumerate(frames):
        f.tofile(str(tmp_path / f"in{k}.bin"))
    r = subprocess.run([exe, str(hdr), str(H), str(W), str(n), str(tmp_path / "in"), str(tmp_path / "out"), str(target), str(R)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in r.stdout.splitlines()]
    assert ["throws_on_bad_scalar", "1"] in lines and ["ofc_calc_time_positive", "1"] in lines
    lines = [l for l in lines if l and l[0] == "out"]
    g = oracle.make_geom(hdr, H, W)
    dt = np.uint16 if hdr else np.uint8
    plan = BlendSchedule(SOURCE_24, target).plan(n)
    flows = {}
    out_index = 0
    for k in range(n):
        count = k + 1                                   # m_frameCount after updateFrame
        if count >= 3:
            flows[k] = oracle.calculate_optical_flow(frames[k - 1], frames[k], g, R)
        for t in plan[k]:
            got = np.fromfile(str(tmp_path / f"out{out_index}.bin"), dtype=dt)
            if count >= 3:
                # warp uses frames N-2, N-1 and the PREVIOUS flow (opticalFlowCalcSDR.cpp:154-156); on the very
                # first flow calc the previous-flow buffer is still zero
                prev = flows[k - 1][1] if (k - 1) in flows else np.zeros((2, g.lh, g.lw), np.int16)
                ref = oracle.warp_frames(frames[k - 2], frames[k - 1], prev, g, np.float32(t), 2)
                assert (got == ref).all(), f"output {out_index} (source frame {k}, t={t})"
                assert ["out", str(out_index), "warp"] == lines[out_index][:3]
                assert int(lines[out_index][5]) == flows[k][2]          # m_totalFrameDelta of the newest flow
            else:
                idx = 0 if count >= 3 else 1 if count >= 2 else 2           # opticalFlowCalcSDR.cpp:173
                src = [frames[max(k - 2, 0)], frames[max(k - 1, 0)], frames[k]][idx] if count >= 3 else (frames[k - 1] if count == 2 else frames[k])
                assert (got == oracle.copy_frame(src, g)).all(), f"copy output {out_index}"
            out_index += 1
