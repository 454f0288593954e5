import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hopperrender_amd import synth
from hopperrender_amd.calc import OpticalFlowCalcSDR, OpticalFlowCalcHDR
from oracle import oracle
import importlib.util
spec = importlib.util.spec_from_file_location('t', os.path.join(os.path.dirname(__file__), '..', 'tests', 'test_random_gpu.py'))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
k = m._case(int(sys.argv[1]))
print(k)
hdr, H, W = k["hdr"], k["H"], k["W"]
rng = np.random.default_rng(5000 + k["seed"])
sc = synth.Scene(H, W, bool(hdr), seed=300 + k["seed"], in_stride=k["si"], max_rect_speed=int(rng.integers(2, 40)))
f = [sc.frame(i) for i in range(4)]
g = oracle.make_geom(hdr, H, W, k["si"], k["so"], k["max_res"])
print("grid", g.lw, g.lh, "rs", g.rs)
cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
for it in range(1, 12):
    for nb in ([k["nb"]] if len(sys.argv) < 3 else [0, k["nb"]]):
        c = cls(H, W, k["si"], k["so"], k["delta"], nb, k["black"], k["white"], k["max_res"], iterations=it, blur_radius=k["blur"], search_radius=k["R"])
        for x in f[:3]: c.updateFrame(x)
        c.calculateOpticalFlow()
        off_a, blur_a, tot_a, oob = oracle.calculate_optical_flow(f[1], f[2], g, k["R"], it, k["delta"], nb, k["blur"])
        off = c.readOffsets()
        bad = (off != off_a)
        print(f"iterations={it} nb={nb} oob={oob} mismatches x={int(bad[0].sum())} y={int(bad[1].sum())} tot {c.m_totalFrameDelta} vs {tot_a}", ("first bad at " + str(np.argwhere(bad)[0])) if bad.any() else "")
        c.close()
