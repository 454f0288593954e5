"""Command-line front end: interpolate a raw NV12 / P010 clip with the filter's own call protocol
(raw-file reader/writer around the caller-protocol replay).

    python -m hopperrender_amd.cli in.nv12 out.nv12 --width 1920 --height 1080 [--hdr] [--target-fps 60]
        [--source-fps 23.976] [--mode 2] [--radius 16] [--scene-threshold 200]

Input: contiguous frames, Y plane then interleaved UV (8-bit NV12, or 16-bit little-endian P010 with --hdr).
Output: the frames the DirectShow filter would deliver, in order (first two periods are copies,
reference HopperRender.cpp:955,1179), two source frames late like the filter (`:940`).
"""
import argparse
import sys

import numpy as np

from .calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
from .protocol import FilterReplay


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input"); ap.add_argument("output")
    ap.add_argument("--width", type=int, required=True); ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--hdr", action="store_true")
    ap.add_argument("--source-fps", type=float, default=24000 / 1001); ap.add_argument("--target-fps", type=float, default=60.0)
    ap.add_argument("--mode", type=int, default=2, help="frame output mode 0-6 (HopperRender.h:10-18)")
    ap.add_argument("--radius", type=int, default=16); ap.add_argument("--delta", type=int, default=8)
    ap.add_argument("--neighbor", type=int, default=6); ap.add_argument("--black", type=float, default=0.0)
    ap.add_argument("--white", type=float, default=255.0); ap.add_argument("--scene-threshold", type=int, default=200)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    dt = np.uint16 if a.hdr else np.uint8
    n_el = a.width * a.height * 3 // 2
    cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
    calc = cls(a.height, a.width, 0, 0, a.delta, a.neighbor, a.black, a.white, 270, device_index=a.device, search_radius=a.radius)
    replay = FilterReplay(calc, int(round(1e7 / a.source_fps)), int(round(1e7 / a.target_fps)), a.mode, a.scene_threshold)
    n_in = n_out = 0
    with open(a.input, "rb") as fi, open(a.output, "wb") as fo:
        while True:
            buf = fi.read(n_el * np.dtype(dt).itemsize)
            if len(buf) < n_el * np.dtype(dt).itemsize:
                break
            for frame in replay.deliver(np.frombuffer(buf, dtype=dt)):
                fo.write(frame.tobytes())
                n_out += 1
            n_in += 1
    print(f"{n_in} source frames -> {n_out} output frames ({sum(1 for k, _ in replay.log if k == 'copy')} copies)", file=sys.stderr)
    calc.close()


if __name__ == "__main__":
    main()
