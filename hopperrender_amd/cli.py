"""Command-line front end: interpolate a raw NV12 / P010 clip with the filter's own call protocol
(raw-file reader/writer around the caller-protocol replay).

    python -m hopperrender_amd.cli in.nv12 out.nv12 --width 1920 --height 1080 [--hdr] [--target-fps 60]
        [--source-fps 23.976] [--mode 2] [--radius 16] [--scene-threshold 200]

    python -m hopperrender_amd.cli in.y4m out.y4m [--target-fps 60]      (size, bit depth and source rate from the header)

Input: contiguous frames, Y plane then interleaved UV (8-bit NV12, or 16-bit little-endian P010 with --hdr), or a
YUV4MPEG2 stream (C420* / C420p10, see y4m.py) when the name ends in .y4m.
Output: the frames the DirectShow filter would deliver, in order (first two periods are copies,
reference HopperRender.cpp:955,1179), two source frames late like the filter (`:940`).
"""
import argparse
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
    a = ap.parse_args(argv)
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
