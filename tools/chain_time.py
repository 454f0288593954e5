"""tools/chain_time.py -- device time of the flow chain alone (one stream, HF_FLAG_PROFILE events), optionally batched."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR

ap = argparse.ArgumentParser()
ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--n", type=int, default=100); ap.add_argument("--radius", type=int, default=16)
ap.add_argument("--batch", type=int, nargs="*", default=[1])
ap.add_argument("--scene", default="bench", choices=synth.SCENES); ap.add_argument("--no-reuse", action="store_true")
a = ap.parse_args()
cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
sc = synth.ContentScene(a.scene, a.H, a.W, bool(a.hdr), 1234)
frames = [sc.frame(k) for k in range(3)]
for B in a.batch:
    cs = [cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=a.radius, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE | (capi.HF_FLAG_NO_SAD_REUSE if a.no_reuse else 0)) for _ in range(B)]
    for c in cs:
        for f in frames: c.updateFrame(f)
    if B == 1:
        run = cs[0].calculateOpticalFlow
    else:
        fb = FlowBatch(cs); run = fb.calculateOpticalFlow
    for _ in range(5): run()
    cs[0].sync(); cs[0].resetProfile()
    t0 = time.perf_counter()
    for _ in range(a.n): run()
    cs[0].sync()
    wall = (time.perf_counter() - t0) / a.n * 1e6
    p = cs[0].profile()
    per_launch = 1e3 * p["flow_ms"] / (p["flow_chains"] / B)
    print(f"flow chain {a.W}x{a.H} hdr={a.hdr} R={a.radius} scene={a.scene} reuse={int(not a.no_reuse)} batch={B}: {per_launch:.2f} us per batched chain = {per_launch / B:.2f} us per pair (wall {wall / B:.1f} us per pair)")
    if B > 1: fb.close()
    for c in cs: c.close()
