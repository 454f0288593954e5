import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hopperrender_amd import synth
from hopperrender_amd.calc import OpticalFlowCalcSDR, OpticalFlowCalcHDR
from oracle import oracle
for hdr, H, W in [(0, 4, 64), (0, 4, 4), (1, 4, 4)]:
    f = [synth.random_frame(H, W, bool(hdr), seed=900 + 7 * i + H + W) for i in range(4)]
    try:
        s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, 0.0, 255.0, 270)
        s.radius(5)
        for x in f[:3]: s.update(x)
        s.calc(); s.update(f[3]); s.calc()
        s.dump_blurred(0, "flow")
        for m in (0, 2):
            s.warp(0.43, m); s.download(f"w{m}")
        js, ref = s.run()
    except Exception as e:
        print(hdr, H, W, "reference failed:", repr(e)[:200]); continue
    c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=5)
    for x in f[:3]: c.updateFrame(x)
    c.calculateOpticalFlow(); c.updateFrame(f[3]); c.calculateOpticalFlow()
    print(hdr, H, W, "flow equal:", (c.readBlurredFlow(0) == ref["flow"]).all())
    g = oracle.make_geom(hdr, H, W)
    for m in (0, 2):
        c.warpFrames(0.43, m)
        hip = c.downloadFrame()
        orc = oracle.warp_frames(f[1], f[2], ref["flow"], g, 0.43, m)
        print("   mode", m, "hip==ref", (hip == ref[f"w{m}"]).all(), " oracle==ref", (orc == ref[f"w{m}"]).all())
hdr, H, W = 0, 4, 64
f = [synth.random_frame(H, W, bool(hdr), seed=900 + 7 * i + H + W) for i in range(4)]
g = oracle.make_geom(hdr, H, W)
for pair in ((0, 1), (1, 2), (2, 3)):
    off, blur, tot, oob = oracle.calculate_optical_flow(f[pair[0]], f[pair[1]], g, 5)
    print("pair", pair, "oracle oob", oob, "max |off|", int(np.abs(off).max()))
s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, 0.0, 255.0, 270); s.radius(5)
for x in f[:3]: s.update(x)
s.calc(); s.dump_offsets("off"); s.dump_blurred(1, "blur")
js, ref = s.run()
c = OpticalFlowCalcSDR(H, W, search_radius=5)
for x in f[:3]: c.updateFrame(x)
c.calculateOpticalFlow()
off, blur, tot, oob = oracle.calculate_optical_flow(f[1], f[2], g, 5)
print("first calc: hip==ref off", (c.readOffsets() == ref["off"]).all(), "oracle==ref off", (off == ref["off"]).all(), "hip==oracle", (c.readOffsets() == off).all(), "oob", oob)
print("blur: hip==ref", (c.readBlurredFlow(1) == ref["blur"]).all(), "oracle==ref", (blur == ref["blur"]).all())
print(g.lw, g.lh, g.rs)
