"""host-side cost of the three batched calls, bench-like: 2 batches of n members (1080p SDR), per-member source frames and plans"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcSDR
from hopperrender_amd.protocol import SOURCE_24, TARGET_60, BlendSchedule
sc = synth.Scene(1080, 1920, False, 7)
dev = []
for k in range(6):
    f = sc.frame(k); d = DeviceBuffer(f.nbytes); d.upload(f); dev.append(d)
plan = BlendSchedule(SOURCE_24, TARGET_60).plan(200)[3:]
for n in (16, 17, 32):
    cs = [OpticalFlowCalcSDR(1080, 1920, search_radius=16, flags=capi.HF_FLAG_ASYNC) for _ in range(2 * n)]
    for s, c in enumerate(cs):
        for k in range(3): c.updateFrameDeviceRef(dev[(s + k) % 6].ptr)
        c.calculateOpticalFlow(); c.sync()
    bs = [FlowBatch(cs[:n]), FlowBatch(cs[n:])]
    outs = [[DeviceBuffer(cs[0].output_frame_bytes) for _ in range(3)] for _ in range(2 * n)]
    op = [[o.ptr for o in outs[i]] for i in range(2 * n)]
    T = [0.0, 0.0, 0.0]; N = 40
    for it in range(N):
        for bi, b in enumerate(bs):
            lo = bi * n
            t0 = time.perf_counter(); b.updateFramesDeviceRef([dev[(s + 3 + it) % 6].ptr for s in range(lo, lo + n)])
            t1 = time.perf_counter(); b.calculateOpticalFlow()
            t2 = time.perf_counter(); b.interpolatePeriod([plan[it] for s in range(lo, lo + n)], op[lo:lo + n], 2)
            t3 = time.perf_counter()
            T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
    for c in cs: c.sync()
    print("n=%2d  update %.1f us  flow %.1f us  period %.1f us  (per batch call; %d periods without a sync)" % (n, 1e6 * T[0] / N / 2, 1e6 * T[1] / N / 2, 1e6 * T[2] / N / 2, N))
    for b in bs: b.close()
    for c in cs: c.close()
