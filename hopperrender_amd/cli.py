"""Command-line front end: interpolate a raw NV12 / P010 clip with the filter's own call protocol
(raw-file reader/writer around the caller-protocol replay).

    python -m hopperrender_amd.cli in.nv12 out.nv12 --width 1920 --height 1080 [--hdr] [--target-fps 60]
        [--source-fps 23.976] [--mode 2] [--radius 16] [--scene-threshold 200]

    python -m hopperrender_amd.cli in.y4m out.y4m [--target-fps 60]      (size, bit depth and source rate from the header)

    python -m hopperrender_amd.cli in.nv12 out.nv12 --width 1920 --height 1080 --gpus 8
        one worker process per GPU (started before anything touches a GPU): the clip's source periods are cut into contiguous
        chunks (batch.shard_timeline: 3 + 12 frames of overlap so that ring, previous flow and scene-change history equal the
        sequential run's), every worker streams its chunk through pinned host rings with asynchronous H2D / D2H on side
        streams (hostio.HostIoRunner) and writes its output frames at their final offsets of the output file -- the results
        are gathered in index order by construction, with no exchange between the workers.
        `--gpus 1` runs the same streaming path in one worker; without --gpus the blocking reference protocol is used.

Input: contiguous frames, Y plane then interleaved UV (8-bit NV12, or 16-bit little-endian P010 with --hdr), or a
YUV4MPEG2 stream (C420* / C420p10, see y4m.py) when the name ends in .y4m.
Output: the frames the DirectShow filter would deliver, in order (first two periods are copies,
reference HopperRender.cpp:955,1179), two source frames late like the filter (`:940`).
"""
import argparse
import os
import sys
from fractions import Fraction

import numpy as np

from .calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
from .protocol import FilterReplay
from .y4m import Y4MReader, Y4MWriter


def _raw_frames(fi, n_el, dt):
    nbytes = n_el * np.dtype(dt).itemsize
    while True:
        buf = fi.read(nbytes)
        if len(buf) < nbytes:
            return
        yield np.frombuffer(buf, dtype=dt)


class _Clip:
    """Random access to the frames of a raw NV12 / P010 file or of a .y4m file with plain `FRAME` records."""

    def __init__(self, path, width, height, hdr, source_fps):
        self.f = open(path, "rb")
        self.y4m = path.lower().endswith(".y4m")
        self.extra = ()
        if self.y4m:
            r = Y4MReader(self.f)
            width, height, hdr = r.width, r.height, r.hdr
            source_fps = source_fps or r.fps
            self.extra = r.extra
            self.data0 = self.f.tell()
        else:
            self.data0 = 0
        if not width or not height:
            raise SystemExit("--width/--height are required for raw input")
        self.width, self.height, self.hdr = width, height, hdr
        self.source_fps = source_fps or 24000 / 1001
        self.dt = np.dtype("<u2") if hdr else np.dtype(np.uint8)
        self.n_el = width * height * 3 // 2
        self.frame_bytes = self.n_el * self.dt.itemsize
        self.record = self.frame_bytes + (6 if self.y4m else 0)
        size = os.fstat(self.f.fileno()).st_size - self.data0
        self.n_frames = size // self.record
        if self.y4m and self.n_frames:
            self.f.seek(self.data0)
            if self.f.read(6) != b"FRAME\n":
                raise SystemExit("--gpus needs a .y4m file with plain FRAME records (no per-frame parameters)")

    def read_into(self, k, out):
        """Source frame k as NV12 / P010 into `out` (e.g. a pinned buffer)."""
        self.f.seek(self.data0 + k * self.record + (6 if self.y4m else 0))
        if not self.y4m:
            got = self.f.readinto(memoryview(out).cast("B"))
            if got != self.frame_bytes:
                raise IOError("short read")
            return
        p = np.frombuffer(self.f.read(self.frame_bytes), dtype=self.dt)
        H, W = self.height, self.width
        from .y4m import planar_to_semiplanar
        out[:] = planar_to_semiplanar(p[:H * W].reshape(H, W), p[H * W:H * W * 5 // 4].reshape(H // 2, W // 2),
                                      p[H * W * 5 // 4:].reshape(H // 2, W // 2), self.hdr)


def _y4m_header(clip, target_fps):
    import io
    r = Fraction(target_fps).limit_denominator(1001)
    b = io.BytesIO()
    Y4MWriter(b, clip.width, clip.height, r.numerator, r.denominator, clip.hdr, clip.extra)
    return b.getvalue()


def _worker(a, rank, world):
    """One rank of the multi-GPU run: its chunk of the timeline through the asynchronous host-I/O path."""
    from . import capi
    from .batch import shard_timeline
    from .hostio import HostIoRunner
    from .y4m import semiplanar_to_planar
    clip = _Clip(a.input, a.width, a.height, a.hdr, a.source_fps)
    src_t, tgt_t = int(round(1e7 / clip.source_fps)), int(round(1e7 / a.target_fps))
    chunk = shard_timeline(clip.n_frames, world, rank, src_t, tgt_t)
    out_y4m = a.output.lower().endswith(".y4m")
    head = len(_y4m_header(clip, a.target_fps)) if out_y4m else 0
    record = clip.frame_bytes + (6 if out_y4m else 0)
    n_dev = max(1, capi.load().hf_device_count())
    runner = HostIoRunner(clip.hdr, clip.height, clip.width, device_index=(a.device + rank) % n_dev, delta_scalar=a.delta,
                          neighbor_scalar=a.neighbor, black=a.black, white=a.white, search_radius=a.radius)
    fd = os.open(a.output, os.O_WRONLY)
    copies = 0

    def sink(i, frame, kind):
        nonlocal copies
        copies += kind == "copy"
        off = head + (chunk.first_output + i) * record
        if out_y4m:
            y, u, v = semiplanar_to_planar(frame, clip.height, clip.width, clip.hdr)
            data = b"FRAME\n" + b"".join(np.ascontiguousarray(p, dtype=clip.dt).tobytes() for p in (y, u, v))
        else:
            data = memoryview(frame).cast("B")
        os.pwrite(fd, data, off)

    kinds = runner.run(chunk, clip.read_into, sink, a.mode, a.scene_threshold, src_t, tgt_t) if chunk.n_periods else []
    os.close(fd)
    runner.close()
    print(f"rank {rank}/{world} device {(a.device + rank) % n_dev}: source periods {chunk.first_period}..{chunk.first_period + chunk.n_periods - 1} "
          f"(+{chunk.first_period - chunk.first_frame} warm-up frames) -> {len(kinds)} output frames from #{chunk.first_output} ({copies} copies)", file=sys.stderr)


def _multi_gpu(a, argv):
    """Parent of the multi-GPU run: never touches a GPU itself; sizes the output file, starts one worker per GPU, waits."""
    import subprocess
    from .protocol import BlendSchedule
    clip = _Clip(a.input, a.width, a.height, a.hdr, a.source_fps)
    sched = BlendSchedule(int(round(1e7 / clip.source_fps)), int(round(1e7 / a.target_fps)))
    n_out = 0
    for _ in range(clip.n_frames):
        n = sched.begin_source_frame()
        for _ in range(n):
            sched.next_scalar()
        n_out += n
    out_y4m = a.output.lower().endswith(".y4m")
    with open(a.output, "wb") as fo:
        head = _y4m_header(clip, a.target_fps) if out_y4m else b""
        fo.write(head)
        fo.truncate(len(head) + n_out * (clip.frame_bytes + (6 if out_y4m else 0)))
    args = [x for x in (argv if argv is not None else sys.argv[1:])]
    procs = [subprocess.Popen([sys.executable, "-m", "hopperrender_amd.cli"] + args + ["--rank", str(r), "--world", str(a.gpus)]) for r in range(a.gpus)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"worker exit codes {rcs}")
    print(f"{clip.n_frames} source frames -> {n_out} output frames on {a.gpus} GPU worker(s)", file=sys.stderr)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input"); ap.add_argument("output")
    ap.add_argument("--width", type=int); ap.add_argument("--height", type=int)
    ap.add_argument("--hdr", action="store_true")
    ap.add_argument("--source-fps", type=float, default=None, help="default: 23.976, or the .y4m header's rate"); ap.add_argument("--target-fps", type=float, default=60.0)
    ap.add_argument("--mode", type=int, default=2, help="frame output mode 0-6 (HopperRender.h:10-18)")
    ap.add_argument("--radius", type=int, default=16); ap.add_argument("--delta", type=int, default=8)
    ap.add_argument("--neighbor", type=int, default=6); ap.add_argument("--black", type=float, default=0.0)
    ap.add_argument("--white", type=float, default=255.0); ap.add_argument("--scene-threshold", type=int, default=200)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=0, help="N worker processes, one per GPU, asynchronous host I/O (0: the blocking reference protocol in this process)")
    ap.add_argument("--rank", type=int, default=-1, help=argparse.SUPPRESS); ap.add_argument("--world", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.rank >= 0:
        return _worker(a, a.rank, a.world)
    if a.gpus > 0:
        return _multi_gpu(a, argv)
    with open(a.input, "rb") as fi, open(a.output, "wb") as fo:
        reader = None
        if a.input.lower().endswith(".y4m"):
            reader = Y4MReader(fi)
            a.width, a.height, a.hdr = reader.width, reader.height, reader.hdr
            if a.source_fps is None:
                a.source_fps = reader.fps
        if not a.width or not a.height:
            ap.error("--width/--height are required for raw input")
        if a.source_fps is None:
            a.source_fps = 24000 / 1001
        dt = np.uint16 if a.hdr else np.uint8
        frames = reader if reader else _raw_frames(fi, a.width * a.height * 3 // 2, dt)
        writer = None
        if a.output.lower().endswith(".y4m"):
            r = Fraction(a.target_fps).limit_denominator(1001)
            writer = Y4MWriter(fo, a.width, a.height, r.numerator, r.denominator, a.hdr, reader.extra if reader else ())
        cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
        calc = cls(a.height, a.width, 0, 0, a.delta, a.neighbor, a.black, a.white, 270, device_index=a.device, search_radius=a.radius)
        replay = FilterReplay(calc, int(round(1e7 / a.source_fps)), int(round(1e7 / a.target_fps)), a.mode, a.scene_threshold)
        n_in = n_out = 0
        for src in frames:
            for frame in replay.deliver(src):
                if writer:
                    writer.write(frame)
                else:
                    fo.write(frame.tobytes())
                n_out += 1
            n_in += 1
    print(f"{n_in} source frames -> {n_out} output frames ({sum(1 for k, _ in replay.log if k == 'copy')} copies)", file=sys.stderr)
    calc.close()


if __name__ == "__main__":
    main()
