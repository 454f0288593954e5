"""tools/prof_one.py -- launch ONE kind of kernel a few times (target for rocprofv3 --pmc passes)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
ap = argparse.ArgumentParser()
ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--n", type=int, default=10); ap.add_argument("--what", default="warp"); ap.add_argument("--mode", type=int, default=2)
ap.add_argument("--radius", type=int, default=16)
a = ap.parse_args()
cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
c = cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=a.radius, flags=capi.HF_FLAG_ASYNC)
sc = synth.Scene(a.H, a.W, bool(a.hdr), 1234)
fr = [sc.frame(k) for k in range(4)]
for f in fr[:3]: c.updateFrame(f)
c.calculateOpticalFlow(); c.updateFrame(fr[3]); c.calculateOpticalFlow(); c.sync()
for _ in range(a.n):
    if a.what == "warp": c.warpFrames(0.3996, a.mode)
    elif a.what == "copy": c.copyFrame()
    else: c.calculateOpticalFlow()
c.sync()
