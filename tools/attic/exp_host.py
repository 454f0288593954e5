"""How long does the HOST take to enqueue one source period (interpolatePeriod) -- is the bench host-bound?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, DeviceBuffer
H, W = 2160, 3840
sc = synth.Scene(H, W, True, 1234)
fr = [sc.frame(k) for k in range(4)]
for flags in (capi.HF_FLAG_ASYNC, capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_GRAPH):
    c = OpticalFlowCalcHDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=flags)
    pool = []
    for f in fr:
        b = DeviceBuffer(f.nbytes); b.upload(f); pool.append(b)
    outs = [DeviceBuffer(c.output_frame_bytes) for _ in range(5)]
    ptrs = [o.ptr for o in outs]
    ts = [0.0, 0.2, 0.4, 0.6, 0.8]
    for k in range(3): c.updateFrameDeviceRef(pool[k].ptr)
    for i in range(5): c.interpolatePeriod(pool[i % 4].ptr, ts, ptrs, 2)
    c.sync()
    n = 50
    t0 = time.perf_counter()
    for i in range(n): c.interpolatePeriod(pool[i % 4].ptr, ts, ptrs, 2)
    t1 = time.perf_counter(); c.sync(); t2 = time.perf_counter()
    t3 = time.perf_counter()
    for i in range(n): c.interpolatePeriod(pool[i % 4].ptr, [], [], 2)
    t4 = time.perf_counter(); c.sync(); t5 = time.perf_counter()
    print(f"flags {flags}: chain only: host enqueue {1e6*(t4-t3)/n:.1f} us per period; total incl. GPU {1e6*(t5-t3)/n:.1f} us per period")
    print(f"flags {flags}: host enqueue {1e6*(t1-t0)/n:.1f} us per period; total incl. GPU {1e6*(t2-t0)/n:.1f} us per period")
    c.close()
