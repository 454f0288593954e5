import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, DeviceBuffer
H, W = 2160, 3840
c = OpticalFlowCalcHDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE)
sc = synth.Scene(H, W, True, 1234)
fr = [sc.frame(k) for k in range(4)]
def t(label, tt=0.3996, mode=2):
    for _ in range(5): c.warpFrames(tt, mode)
    c.resetProfile()
    for _ in range(40): c.warpFrames(tt, mode)
    p = c.profile(); print(f"{label:40s} {1e3*p['warp_ms']/p['warp_launches']:.2f} us")
for f in fr[:3]: c.updateFrame(f)
c.calculateOpticalFlow(); c.updateFrame(fr[3]); c.calculateOpticalFlow(); c.sync()
t("distinct frames, real flow")
real = c.readBlurredFlow(0)
c.writeBlurredFlow(0, np.zeros_like(real)); t("distinct frames, zero flow")
db = DeviceBuffer(fr[0].nbytes); db.upload(fr[0])
for _ in range(3): c.updateFrameDeviceRef(db.ptr)
c.writeBlurredFlow(0, real); t("same frame twice, real flow")
c.writeBlurredFlow(0, np.zeros_like(real)); t("same frame twice, zero flow (B hits L1)")
t("same frame, zero flow, mode 0", mode=0)
