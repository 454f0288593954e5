"""tools/r02/warp_only.py [launches] [members] -- nothing but a few fused 2160p HDR period warps (5 outputs, mode 2):
the shortest program for rocprofv3 --pmc passes over warp_fast_kernel.  members > 1: one batched launch per period."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_mem = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = synth.Scene(2160, 3840, True, 1234)
frames = [sc.frame(k) for k in range(3)]
dev = [DeviceBuffer(f.nbytes) for f in frames]
for d, f in zip(dev, frames): d.upload(f)
cs = [OpticalFlowCalcHDR(2160, 3840, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC) for _ in range(n_mem)]
ts = [0.0, 0.1998, 0.3996, 0.5994, 0.7992]
outs = [[DeviceBuffer(cs[0].output_frame_bytes) for _ in range(5)] for _ in range(n_mem)]
if n_mem == 1:
    c = cs[0]
    for d in dev: c.updateFrameDeviceRef(d.ptr)
    c.calculateOpticalFlow(); c.sync()
    for _ in range(n_launch): c.interpolateOnly(ts, [o.ptr for o in outs[0]], 2)
    c.sync()
else:
    b = FlowBatch(cs)
    for d in dev: b.updateFramesDeviceRef([d.ptr] * n_mem)
    b.calculateOpticalFlow()
    for _ in range(n_launch): b.interpolatePeriod([ts] * n_mem, [[o.ptr for o in outs[i]] for i in range(n_mem)], 2)
    for c in cs: c.sync()
print("ok")
