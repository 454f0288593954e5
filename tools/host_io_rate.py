"""tools/host_io_rate.py -- PCIe-INCLUSIVE rates (never bench.py's `value`; bench.py runs this as a child for its `host_io` block):
updateFrame(host) -> calculateOpticalFlow -> per output frame warpFrames + downloadFrame(host), one context."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR, PinnedArray
from hopperrender_amd.protocol import SOURCE_24, TARGET_120, TARGET_60, BlendSchedule
ap = argparse.ArgumentParser(); ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--H", type=int, default=2160)
ap.add_argument("--W", type=int, default=3840); ap.add_argument("--n", type=int, default=30); ap.add_argument("--target", type=int, default=TARGET_120)
ap.add_argument("--device", type=int, default=0); ap.add_argument("--async-only", action="store_true", help="only the asynchronous pinned variant (multi-rank runs)")
a = ap.parse_args()
cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
sc = synth.Scene(a.H, a.W, bool(a.hdr), 1234)
frames = [sc.frame(k) for k in range(4)]
res = {}
for pinned in (() if a.async_only else (False, True)):
    c = cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, device_index=a.device)
    n_el = c.output_frame_bytes // np.dtype(c.dtype).itemsize
    if pinned:
        ins = [PinnedArray(f.size, c.dtype) for f in frames]
        for p, f in zip(ins, frames): p.array[:] = f
        src = [p.array for p in ins]
        outp = PinnedArray(n_el, c.dtype); out = outp.array
    else:
        src = frames; out = np.empty(n_el, c.dtype)
    plan = BlendSchedule(SOURCE_24, a.target).plan(a.n + 4)
    for k in range(3): c.updateFrame(src[k])
    c.calculateOpticalFlow()
    t0 = time.perf_counter(); nout = 0
    for i in range(a.n):
        c.updateFrame(src[i % 4]); c.calculateOpticalFlow()
        for t in plan[i + 3]:
            c.warpFrames(t, 2); c.downloadFrame(out); nout += 1
    dt = time.perf_counter() - t0
    res["blocking_pinned" if pinned else "blocking_pageable"] = {"frames_per_s": round(nout / dt, 1), "d2h_GB_per_s": round(nout * c.output_frame_bytes / dt / 1e9, 2)}
    c.close()
# asynchronous pipeline: uploads/readbacks on side streams, pinned buffers, nothing blocks until the end
from hopperrender_amd import capi
c = cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, device_index=a.device, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM)
n_el = c.output_frame_bytes // np.dtype(c.dtype).itemsize
ins = [PinnedArray(f.size, c.dtype) for f in frames]
for p, f in zip(ins, frames): p.array[:] = f
outs = [PinnedArray(n_el, c.dtype) for _ in range(8)]
plan = BlendSchedule(SOURCE_24, a.target).plan(a.n + 4)
for k in range(3): c.updateFrameAsync(ins[k])
c.calculateOpticalFlow(); c.sync()
t0 = time.perf_counter(); nout = 0
for i in range(a.n):
    c.updateFrameAsync(ins[i % 4]); c.calculateOpticalFlow()
    for t in plan[i + 3]:
        c.warpFrames(t, 2); c.downloadFrameAsync(outs[nout % 8]); nout += 1
    if i % 2 == 1: c.sync()     # bound the number of in-flight host buffers (8 outputs here)
c.sync(); dt = time.perf_counter() - t0
res["async_pinned_side_streams"] = {"frames_per_s": round(nout / dt, 1), "d2h_GB_per_s": round(nout * c.output_frame_bytes / dt / 1e9, 2),
                                    "h2d_GB_per_s": round(a.n * c.input_frame_bytes / dt / 1e9, 2)}
c.close()
print(json.dumps(res))
