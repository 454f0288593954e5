"""Device-memory leak check: contexts, batches (single- and dual-stream members), caller buffers created and destroyed in a loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, FlowBatch, DeviceBuffer
H, W = 1080, 1920
f = synth.random_frame(H, W, True, 1)
def cycle(n):
    for i in range(n):
        cs = [OpticalFlowCalcHDR(H, W, flags=capi.HF_FLAG_ASYNC | (capi.HF_FLAG_DUAL_STREAM if i % 2 else 0), search_radius=8) for _ in range(3)]
        b = FlowBatch(cs)
        for c in cs:
            for _ in range(3): c.updateFrame(f)
        b.calculateOpticalFlow()
        outs = [DeviceBuffer(cs[0].output_frame_bytes) for _ in range(3)]
        for c in cs: c.interpolateOnly([0.2, 0.5, 0.8], [o.ptr for o in outs], 2)
        for c in cs: c.sync()
        b.close()
        for c in cs: c.close()
        for o in outs: o.free()
cycle(3)
torch.cuda.synchronize(); free0, total = torch.cuda.mem_get_info()
cycle(40)
torch.cuda.synchronize(); free1, _ = torch.cuda.mem_get_info()
print("free before %.1f MB, after %.1f MB, delta %.1f MB" % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6))
