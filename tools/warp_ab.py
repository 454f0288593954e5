"""tools/warp_ab.py -- stand-alone times of the fused period warp for ONE build of the library (HF_LIB selects it), plus a
checksum of every output frame so that variants can be compared bit for bit:

    HF_LIB=hopperrender_amd/lib/exp/X/libhopperflow.so python tools/warp_ab.py [--members 16] [--hdr 1 --H 2160 --W 3840]

  single hot / cold   one context's 5-output period (mode 2) per launch, same buffers / rotating buffers (HBM-cold)
  batch               one launch per period for `--members` contexts (hf_batch), sources and outputs rotating
Prints one JSON line."""
import argparse, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import capi, synth
from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR

ap = argparse.ArgumentParser()
ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--members", type=int, default=16); ap.add_argument("--n", type=int, default=30); ap.add_argument("--mode", type=int, default=2)
ap.add_argument("--scene", default="bench", help="bench: the bench's synthetic scene; fast: rectangles up to 160 px per frame")
a = ap.parse_args()
cls = OpticalFlowCalcHDR if a.hdr else OpticalFlowCalcSDR
dt = np.uint16 if a.hdr else np.uint8
sc = synth.Scene(a.H, a.W, bool(a.hdr), 1234, **({"max_rect_speed": 160} if a.scene == "fast" else {}))
fr = [sc.frame(k) for k in range(6)]
ts5 = [0.0, 0.1998, 0.3996, 0.5994, 0.7992]
res = {"lib": os.environ.get("HF_LIB", "default"), "scene": a.scene}

# ---- single context ----
c = cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE)
F = c.output_frame_bytes
K = 6
srcs = [DeviceBuffer(fr[0].nbytes) for _ in range(K + 2)]
for i, b in enumerate(srcs): b.upload(fr[i % 6])
outsK = [[DeviceBuffer(F) for _ in range(5)] for _ in range(K)]
for k in range(4):
    c.updateFrameDeviceRef(srcs[k].ptr)
    if k >= 2: c.calculateOpticalFlow()
c.sync()
h = hashlib.sha256()
c.interpolateOnly(ts5, [o.ptr for o in outsK[0]], a.mode); c.sync()
for o in outsK[0]: h.update(o.download(dt).tobytes())
res["sha_single"] = h.hexdigest()[:16]
for _ in range(3): c.interpolateOnly(ts5, [o.ptr for o in outsK[0]], a.mode)
c.sync(); c.resetProfile()
for _ in range(a.n): c.interpolateOnly(ts5, [o.ptr for o in outsK[0]], a.mode)
p = c.profile(); res["single_hot_us"] = round(1e3 * p["warp_ms"] / p["warp_launches"], 2)
real = c.readBlurredFlow(0)
def cold_pass(n):
    for i in range(n):
        c.updateFrameDeviceRef(srcs[(i + 3) % (K + 2)].ptr)
        c.interpolateOnly(ts5, [o.ptr for o in outsK[i % K]], a.mode)
cold_pass(K); c.sync(); c.resetProfile(); cold_pass(3 * K); c.sync()
p = c.profile(); res["single_cold_us"] = round(1e3 * p["warp_ms"] / p["warp_launches"], 2)
c.close()
for row in outsK:
    for o in row: o.free()

# ---- one batched launch per period ----
n = a.members
if n > 1:
    ms = [cls(a.H, a.W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_PROFILE | capi.HF_FLAG_NO_TIMING) for _ in range(n)]
    b = FlowBatch(ms)
    outs = [[[DeviceBuffer(F) for _ in range(5)] for _ in range(n)] for _ in range(2)]       # two rotating output sets
    plans = [ts5[i % 5:] + ts5[:i % 5] for i in range(n)]
    for k in range(4):
        b.updateFramesDeviceRef([srcs[(k + i) % (K + 2)].ptr for i in range(n)])
        if k >= 2: b.calculateOpticalFlow()
    prep = [b.preparePeriod(None, plans, [[o.ptr for o in outs[s][i]] for i in range(n)], a.mode, calculate_flow=False) for s in range(2)]
    b.runPeriod(prep[0]); b.sync()
    h = hashlib.sha256()
    for i in range(n):
        for o in outs[0][i]: h.update(o.download(dt).tobytes())
    res["sha_batch"] = h.hexdigest()[:16]
    for i in range(4): b.runPeriod(prep[i % 2])
    b.sync(); ms[0].resetProfile()
    for i in range(a.n): b.runPeriod(prep[i % 2])
    b.sync()
    p = ms[0].profile()
    res["batch_launch_us"] = round(1e3 * p["warp_ms"] / p["warp_launches"], 1)
    res["batch_us_per_member"] = round(1e3 * p["warp_ms"] / p["warp_launches"] / n, 2)
    res["batch_real_TBps"] = round(n * 7 * F / (1e-3 * p["warp_ms"] / p["warp_launches"]) / 1e12, 3)
    b.close()
    for m in ms: m.close()
print(json.dumps(res), flush=True)
