"""GPU: the multi-GPU host-I/O driver (reference traffic opticalFlowCalcSDR.cpp:19-42): frames enter
and leave through host memory on asynchronous side streams with pinned rings (hopperrender_amd/hostio.py), the timeline is
cut into one contiguous chunk per rank (batch.shard_timeline) and the output frames are gathered in index order -- every
output frame must equal the sequential, blocking filter replay, cut periods (copyFrame output) included."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cut_clip(H, W, hdr, n_before, n_after, seed=42):
    from hopperrender_amd import synth
    a = synth.Scene(H, W, bool(hdr), seed)
    b = synth.Scene(H, W, bool(hdr), seed + 999)
    return [a.frame(k) for k in range(n_before)] + [b.frame(n_before + k) for k in range(n_after)]


def sequential(frames, H, W, hdr, target, thr, R=8):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, FilterReplay
    seq = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=R)
    replay = FilterReplay(seq, SOURCE_24, target, scene_change_threshold=thr)
    want = []
    for f in frames:
        want += [o.copy() for o in replay.deliver(f)]
    kinds = [k for k, _ in replay.log]
    seq.close()
    return want, kinds


@pytest.mark.parametrize("hdr,world,cut_at,out_ring", [(0, 3, 20, 12), (1, 2, 27, 2), (0, 1, 14, 5), (0, 4, 30, 3)])
def test_async_host_io_chunks_equal_the_sequential_filter(native_lib, hdr, world, cut_at, out_ring):
    """Every rank's chunk through HostIoRunner (pinned rings, hf_update_frame_async / hf_download_frame_async, one hf_wait_flow
    per period), concatenated in rank order == the blocking sequential replay; also with an output ring so small that slots
    are drained and reused all the time."""
    from hopperrender_amd import batch
    from hopperrender_amd.hostio import HostIoRunner
    from hopperrender_amd.protocol import SOURCE_24, TARGET_60, TARGET_120
    H, W, n, thr = 180, 320, 40, 150
    target = TARGET_120 if hdr else TARGET_60
    frames = cut_clip(H, W, hdr, cut_at, n - cut_at, seed=5)
    want, want_kinds = sequential(frames, H, W, hdr, target, thr)
    assert "copy" in want_kinds[6:], "the clip must contain a detected scene change"
    got, got_kinds = [], []
    for rank in range(world):
        chunk = batch.shard_timeline(n, world, rank, SOURCE_24, target)
        r = HostIoRunner(hdr, H, W, search_radius=8, out_ring=out_ring)
        seen = []

        def fill(k, arr):
            arr[:] = frames[k]

        def sink(i, arr, kind):
            assert i == len(seen)                          # strictly in order
            seen.append(arr.copy())

        kinds = r.run(chunk, fill, sink, 2, thr, SOURCE_24, target)
        assert chunk.first_output == len(got) and len(seen) == len(kinds)
        assert r.bytes_in == chunk.n_frames * frames[0].nbytes and r.bytes_out == len(seen) * frames[0].nbytes
        got += seen; got_kinds += kinds
        r.close()
    assert got_kinds == want_kinds
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert (a == b).all(), i


@pytest.mark.parametrize("ext,hdr", [("nv12", 0), ("y4m", 1)])
def test_cli_two_gpu_workers_equal_the_sequential_cli(native_lib, tmp_path, ext, hdr):
    """`python -m hopperrender_amd.cli --gpus 2`: two worker PROCESSES (both on the one GPU of this box), each streaming its
    chunk through host memory and writing its output frames at their final file offsets == the single-process blocking CLI,
    byte for byte, for a clip with a hard cut (raw NV12 and 10-bit .y4m)."""
    from hopperrender_amd.y4m import Y4MWriter
    H, W, n, cut_at = 180, 320, 36, 19
    frames = cut_clip(H, W, hdr, cut_at, n - cut_at, seed=9)
    src = tmp_path / f"in.{ext}"
    with open(src, "wb") as f:
        if ext == "y4m":
            w = Y4MWriter(f, W, H, 24000, 1001, bool(hdr))
            for x in frames:
                w.write(x)
        else:
            for x in frames:
                f.write(x.tobytes())
    common = [str(src)]
    geo = [] if ext == "y4m" else ["--width", str(W), "--height", str(H)]
    opts = geo + ["--radius", "8", "--scene-threshold", "150", "--target-fps", "60"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(out, extra):
        r = subprocess.run([sys.executable, "-m", "hopperrender_amd.cli", str(src), str(out)] + opts + extra, capture_output=True, text=True,
                           env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stderr

    log_seq = run(tmp_path / f"seq.{ext}", [])
    log_two = run(tmp_path / f"two.{ext}", ["--gpus", "2"])
    log_one = run(tmp_path / f"one.{ext}", ["--gpus", "1"])
    a = open(tmp_path / f"seq.{ext}", "rb").read()
    assert len(a) > 0 and "copies" in log_seq
    assert open(tmp_path / f"two.{ext}", "rb").read() == a, log_two
    assert open(tmp_path / f"one.{ext}", "rb").read() == a, log_one
    assert "rank 0/2" in log_two and "rank 1/2" in log_two
    n_copies = sum(int(l.split("(")[-1].split()[0]) for l in log_two.splitlines() if l.startswith("rank"))
    assert n_copies > 5          # the two start-up periods of rank 0 (HopperRender.cpp:955) + the cut period


def test_c_example_two_workers_equal_the_sequential_cli(native_lib, tmp_path):
    """examples/hf_interpolate_clip.c -- plain C11 against include/hopperflow.h only: hf_shard_timeline + hf_hostio_run, one worker
    process per GPU (fork + exec before any GPU call), pread into / pwrite out of page-locked buffers -- produces the same bytes
    as the sequential blocking CLI for a P010 clip with a hard cut (both workers on the one GPU of this box)."""
    from hopperrender_amd import build
    H, W, n, cut_at = 180, 320, 36, 17
    frames = cut_clip(H, W, 1, cut_at, n - cut_at, seed=11)
    src = tmp_path / "in.p010"
    with open(src, "wb") as f:
        for x in frames:
            f.write(x.tobytes())
    exe = tmp_path / "hf_interpolate_clip"
    lib = os.path.dirname(build.LIB_FLOW)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "hf_interpolate_clip.c"), "-L", lib, "-lhopperflow", "-Wl,-rpath," + lib, "-o", str(exe)])
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "hopperrender_amd.cli", str(src), str(tmp_path / "seq.p010"), "--width", str(W), "--height", str(H), "--hdr",
                        "--radius", "8", "--scene-threshold", "150", "--target-fps", "120"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    want = open(tmp_path / "seq.p010", "rb").read()
    for gpus in (2, 3):
        out = tmp_path / f"c{gpus}.p010"
        r = subprocess.run([str(exe), str(src), str(out), str(W), str(H), "1", "120", str(gpus), "8", "150"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(out, "rb").read() == want, r.stderr[-2000:]
        assert f"rank {gpus - 1}/{gpus}" in r.stderr and "copies" in r.stderr


def test_wait_flow_and_wait_download(native_lib):
    """hf_wait_flow: m_totalFrameDelta of the chain just enqueued without draining the side streams; hf_wait_download: per
    readback completion in issue order."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import OpticalFlowCalcSDR, PinnedArray
    H, W = 360, 640
    sc = synth.Scene(H, W, False, 3)
    fr = [sc.frame(k) for k in range(5)]
    ref = OpticalFlowCalcSDR(H, W, search_radius=9)
    c = OpticalFlowCalcSDR(H, W, search_radius=9, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM)
    pins = [PinnedArray(f.size, np.uint8) for f in fr]
    for p, f in zip(pins, fr):
        p.array[:] = f
    outs = [PinnedArray(f.size, np.uint8) for _ in range(4)]
    for k in range(5):
        ref.updateFrame(fr[k]); c.updateFrameAsync(pins[k])
        if k < 2:
            continue
        ref.calculateOpticalFlow(); c.calculateOpticalFlow()
        c.waitFlow()
        assert c.m_totalFrameDelta == ref.m_totalFrameDelta, k
    assert c.downloadsIssued() == 0
    with pytest.raises(capi.HopperFlowError):
        c.waitDownload(0)
    want = []
    for i, t in enumerate([0.0, 0.25, 0.5, 0.75]):
        ref.warpFrames(t, 2); want.append(ref.downloadFrame().copy())
        c.warpFrames(t, 2); c.downloadFrameAsync(outs[i])
    assert c.downloadsIssued() == 4
    for i in (3, 0, 2, 1):          # any order: an in-order stream, later events imply earlier ones
        c.waitDownload(i)
        assert (outs[i].array == want[i]).all(), i
    c.sync()
    # more readbacks in flight than the context keeps events for (64): waiting for an old index must still mean "it has landed"
    many = [PinnedArray(fr[0].size, np.uint8) for _ in range(80)]
    ts = [i / 100.0 for i in range(80)]
    base = c.downloadsIssued()
    for i, t in enumerate(ts):
        c.warpFrames(t, 2); c.downloadFrameAsync(many[i])
    assert c.downloadsIssued() == base + 80
    for i in (0, 7, 15, 16, 63, 79):
        c.waitDownload(base + i)
        ref.warpFrames(ts[i], 2)
        assert (many[i].array == ref.downloadFrame()).all(), i
    c.sync()
    ref.close(); c.close()
    for p in pins + outs + many:
        p.free()
