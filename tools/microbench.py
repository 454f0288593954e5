"""tools/microbench.py -- per-kernel device times on one stream (HIP events via HF_FLAG_PROFILE)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR

ap = argparse.ArgumentParser()
ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--n", type=int, default=50); ap.add_argument("--radius", type=int, default=16)
a = ap.parse_args()
cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
c = cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=a.radius, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE)
sc = synth.Scene(a.H, a.W, bool(a.hdr), 1234)
fr = [sc.frame(k) for k in range(4)]
for f in fr[:3]: c.updateFrame(f)
c.calculateOpticalFlow(); c.updateFrame(fr[3]); c.calculateOpticalFlow(); c.sync()
F = c.output_frame_bytes
def t_warp(label, mode, t):
    for _ in range(5): c.warpFrames(t, mode)
    c.resetProfile()
    for _ in range(a.n): c.warpFrames(t, mode)
    p = c.profile(); us = 1e3 * p["warp_ms"] / p["warp_launches"]
    nb = {0: 2, 1: 2, 2: 3}.get(mode, 3) * F
    print(f"{label:34s} {us:8.2f} us  {nb/us/1e3:8.1f} GB/s (algorithmic {nb/1e6:.1f} MB)")
def t_copy():
    for _ in range(5): c.copyFrame()
    c.resetProfile()
    for _ in range(a.n): c.copyFrame()
    p = c.profile(); us = 1e3 * p["copy_ms"] / p["copy_launches"]
    print(f"{'copy':34s} {us:8.2f} us  {2*F/us/1e3:8.1f} GB/s")
def t_flow():
    for _ in range(3): c.calculateOpticalFlow()
    c.resetProfile()
    for _ in range(a.n): c.calculateOpticalFlow()
    p = c.profile(); print(f"{'flow chain R=%d' % a.radius:34s} {1e3*p['flow_ms']/p['flow_chains']:8.2f} us")
t_copy()
for m in (0, 1, 2): t_warp(f"warp mode {m} real flow t=0.3996", m, 0.3996)
real = c.readBlurredFlow(0)
c.writeBlurredFlow(0, np.zeros_like(real)); t_warp("warp mode 2 zero flow", 2, 0.3996); t_warp("warp mode 0 zero flow", 0, 0.3996)
z = np.zeros_like(real); z[0] = 17; z[1] = -9
c.writeBlurredFlow(0, z); t_warp("warp mode 2 const flow (17,-9)", 2, 0.3996); 
z[0] = 16; z[1] = -8
c.writeBlurredFlow(0, z); t_warp("warp mode 2 const flow (16,-8) t=.5", 2, 0.5)
c.writeBlurredFlow(0, real)
t_flow()
# update (device-resident): D2D copy + prep_phase
from hopperrender_amd.calc import DeviceBuffer
db = DeviceBuffer(fr[0].nbytes); db.upload(fr[0])
for _ in range(3): c.updateFrameDevice(db.ptr)
c.sync(); c.timerBegin()
for _ in range(a.n): c.updateFrameDevice(db.ptr)
print(f"{'updateFrameDevice (D2D + prep)':34s} {1e3*c.timerEnd()/a.n:8.2f} us")
# fused period: 5 outputs in one launch
outs5 = [DeviceBuffer(c.output_frame_bytes) for _ in range(5)]
ptrs5 = [o.ptr for o in outs5]
ts5 = [0.0, 0.1998, 0.3996, 0.5994, 0.7992]
for _ in range(3): c.interpolateOnly(ts5, ptrs5, 2)
c.resetProfile()
for _ in range(a.n): c.interpolateOnly(ts5, ptrs5, 2)
p = c.profile(); us = 1e3 * p["warp_ms"] / p["warp_launches"]
print(f"{'fused period (5 outputs, mode 2)':34s} {us:8.2f} us  {5*3*F/us/1e3:8.1f} GB/s algorithmic, {7*F/us/1e3:8.1f} GB/s compulsory (2F + 5F)")
# the same fused period HBM-cold: rotate over enough source frames / output sets that nothing survives in the 256 MB MALL
K = 6
srcs = [DeviceBuffer(fr[0].nbytes) for _ in range(K + 2)]
for i, b in enumerate(srcs): b.upload(fr[i % 4])
outsK = [[DeviceBuffer(c.output_frame_bytes) for _ in range(5)] for _ in range(K)]
real = c.readBlurredFlow(0)
for k in range(3): c.updateFrameDeviceRef(srcs[k].ptr)
c.sync(); c.writeBlurredFlow(0, real)
def cold_pass(n, mode):
    for i in range(n):
        c.updateFrameDeviceRef(srcs[(i + 3) % (K + 2)].ptr)      # rotates the ring (prep kernel runs too; not in the warp span)
        c.interpolateOnly(ts5, [o.ptr for o in outsK[i % K]], mode)
for mode in (2, 0):
    cold_pass(K, mode); c.sync(); c.resetProfile()
    cold_pass(3 * K, mode); c.sync()
    p = c.profile(); us = 1e3 * p["warp_ms"] / p["warp_launches"]
    nsrc = 2 if mode == 2 else 1
    print(f"{'fused period mode %d, HBM-cold' % mode:34s} {us:8.2f} us  {(nsrc+5)*F/us/1e3:8.1f} GB/s compulsory ({nsrc}F + 5F)")
