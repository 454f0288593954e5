import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hopperrender_amd import synth, capi
from hopperrender_amd.calc import OpticalFlowCalcSDR
from oracle import oracle
H, W = 1080, 1920
sc = synth.ContentScene("bench", H, W, False, 1234)
f = [sc.frame(i) for i in range(3)]
g = oracle.make_geom(0, H, W)
for it in (4, 5, 6, 7, 8):
    c = OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, iterations=it)
    for x in f: c.updateFrame(x)
    c.calculateOpticalFlow(); c.sync()
    off, blur, tot, _ = oracle.calculate_optical_flow(f[1], f[2], g, 16, it, 8, 6, 4)
    o = c.readOffsets()
    bad = (o != off)
    print("iterations", it, "mismatches X", int(bad[0].sum()), "Y", int(bad[1].sum()))
    if bad.any():
        ys, xs = np.nonzero(bad[0] | bad[1])
        print("  first bad px", xs[:6], ys[:6], "gpu", o[:, ys[0], xs[0]], "ref", off[:, ys[0], xs[0]])
        ws = 256 >> (it - 1)
        wxs, wys = np.unique(xs // ws), np.unique(ys // ws)
        print("  bad windows:", len(set(zip((xs // ws).tolist(), (ys // ws).tolist()))), "of", (480 // ws) * (270 // ws + 1), " wy range", wys.min(), wys.max(), "wx range", wxs.min(), wxs.max())
    c.close()
