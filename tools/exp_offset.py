import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, DeviceBuffer
H, W = 2160, 3840
c = OpticalFlowCalcHDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE)
sc = synth.Scene(H, W, True, 1234)
fr = [sc.frame(k) for k in range(4)]
for f in fr[:3]: c.updateFrame(f)
c.calculateOpticalFlow(); c.updateFrame(fr[3]); c.calculateOpticalFlow(); c.sync()
big = DeviceBuffer(c.output_frame_bytes + (8 << 20))
for off in (0, 256, 1024, 4096 + 256, 65536 + 512, (1 << 20) + 768, (3 << 20) + 16 * 1001):
    c.setOutputBuffer(big.ptr + off)
    for _ in range(5): c.warpFrames(0.3996, 2)
    c.resetProfile()
    for _ in range(40): c.warpFrames(0.3996, 2)
    p = c.profile()
    print(f"out offset {off:9d}: {1e3*p['warp_ms']/p['warp_launches']:.2f} us")
