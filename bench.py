#!/usr/bin/env python3
"""bench.py -- interpolated frames/s (+ ms per flow calc) of the HIP OpticalFlowCalc path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hdr2160_24to120] [--radius 16]
    (N > 1: launched by torch.distributed.run, one rank per GPU)

A STEP = `--periods-per-step` (64) consecutive source-frame periods for every one of the `--streams` independent
frame-pair streams a rank owns.  One period of one stream = updateFrame of a device-resident source frame (zero-copy
reference into the 3-frame ring + phase-plane build), calculateOpticalFlow (16-step refinement + blur), then the
warpFrames(t, BlendedFrame) calls the reference filter would issue for that period (24->120 fps: 5 or 6 outputs,
reference HopperRender.cpp:944-948,1192-1197), each into its own output frame in HBM.  Streams are grouped into
batches (hf_batch): one phase-plane launch, one batched refinement chain and ONE fused warp launch per batch and period.
Streams are independent (flow has no temporal state), so ranks shard them with NO collective:
weak scaling.

One JSON line on rank 0:
  value              whole-job interpolated frames/s, inputs and outputs resident in HBM
  roofline           the HBM roofline of the PIPELINE as SURVEY.md 8(d) defines it: frac = value x B_out / 8 TB/s,
                     B_out = 3F + 4N algorithmic bytes per output frame; plus the dominant kernel (the fused period warp)
                     alone on the GPU, priced both with the algorithmic bytes it stands for and with its real HBM traffic
                     (profiles/roofline_traffic.json, generated from rocprofv3 PMC passes by tools/pmc_traffic.py), and its
                     launch duration inside the timed region (HIP events of the dispatch itself)
  host_io            the same path with host buffers: pinned asynchronous H2D / D2H on side streams (PCIe-inclusive,
                     never `value`)
  cpu_baseline       the plain-C oracle port on a bounded sample, 1 thread and all host cores
  reference_opencl   the reference's own OpenCL path (oracle/_ref) on the same GPU when it is available
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of which the
# null stream takes): two HIP streams that share a queue run strictly one after the other, and with 6 or more queues in
# use the device gets slower again (DESIGN.md "Hardware queues").  So: 1 (null stream) + one queue per batch stream,
# at most 5.  Must be set before the HIP runtime initialises; an explicit setting by the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")

WORKLOADS = {
    # name: (hdr, H, W, target frame time in 100ns units, description)   source = 23.976 fps
    "hdr2160_24to120": (1, 2160, 3840, 83333, "3840x2160 HDR (P010), 24->120 fps, R=16, full pyramid, blend (BASELINE config 3)"),
    "sdr1080_24to60": (0, 1080, 1920, 166667, "1920x1080 SDR (NV12), 24->60 fps, R=16, full pyramid, blend (BASELINE config 2)"),
    "hdr2160_24to60": (1, 2160, 3840, 166667, "3840x2160 HDR (P010), 24->60 fps"),
    "sdr1080_24to120": (0, 1080, 1920, 83333, "1920x1080 SDR (NV12), 24->120 fps"),
    "sdr360_24to60": (0, 360, 640, 166667, "640x360 SDR, 24->60 fps (plumbing size)"),
    # north_star: "synthetic 1080p/2160p SDR+HDR pairs" -- the two size / depth combinations BASELINE's configs do not name
    "hdr1080_24to120": (1, 1080, 1920, 83333, "1920x1080 HDR (P010), 24->120 fps, R=16, full pyramid, blend"),
    "sdr2160_24to60": (0, 2160, 3840, 166667, "3840x2160 SDR (NV12), 24->60 fps, R=16, full pyramid, blend"),
    # BASELINE config 4: 64 independent 1080p SDR frame pairs in flight, sharded over the ranks (64 / world pair streams per GPU,
    # one hf_batch per GPU up to 32 members); BASELINE config 5: 2160p HDR with the neighbour scalar and the blur radius turned up
    "sdr1080_64pairs": (0, 1080, 1920, 166667, "1920x1080 SDR, 64 independent frame pairs sharded across the GPUs, 24->60 fps (BASELINE config 4)"),
    "hdr2160_nb10_blur32": (1, 2160, 3840, 83333, "3840x2160 HDR, neighbor scalar 10, blurFlow radius 32, 24->120 fps (BASELINE config 5)"),
}
# per workload: (pair streams per GPU, pairs per flow batch) -- measured operating points (tools/scan_op.sh).  Round 4:
# FOUR batch streams of 12 instead of two of 16 at 2160p (+ 6.5 %: 84.8-85.4 k against 79.4-80.2 k frames/s on one box; 3 x 12 + 4.3 %, 4 x 16
# + 2.3 %, 5 or 6 streams and batches of 13-14 lose), three of 12 at 1080p (+ 4 %) -- with the faster chain and warp launches of this round
# more, smaller batches in flight fill the device better; four is also the number of hardware queues the batch streams can have.
OPERATING_POINT = {"hdr2160_24to120": (48, 12), "hdr2160_24to60": (48, 12), "sdr1080_24to60": (36, 12), "sdr1080_24to120": (36, 12),
                   "sdr360_24to60": (32, 16), "hdr2160_nb10_blur32": (48, 12), "hdr1080_24to120": (36, 12), "sdr2160_24to60": (48, 12)}
WORKLOAD_PARAMS = {"hdr2160_nb10_blur32": {"neighbor": 10, "blur_radius": 32}}   # overrides of --neighbor / --blur-radius
TOTAL_PAIRS = {"sdr1080_64pairs": 64}                                            # pair streams of the whole JOB (strong-scaled over ranks)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# the dominant kernel of the batched pipeline: 2160p HDR (one flow cell per 16-byte thread) runs the LDS-staged period warp
# (one window per workgroup of 4 wave tiles), 1080p SDR the global-path kernel
# (template prefix: the full symbol -- waves per workgroup, rows per thread -- is read from the shipped binary, see warp_symbol())
# keyed by (hdr, frame larger than 1080p): 2160p HDR (flow cell = one 16-byte thread) takes the staged period warp, the others the global-path
# kernel of their element size and cell width (names as the PMC passes of profiles/roofline_traffic.json saw them dispatched)
WARP_SYMBOL_PREFIX = {(1, True): "warp_wg_kernel<unsigned short, 2,", (0, False): "warp_fast_kernel<unsigned char, 4, 2, 2, 8, true>",
                      (0, True): "warp_fast_kernel<unsigned char, 8, 2, 2, 16, true>", (1, False): "warp_fast_kernel<unsigned short, 4, 2, 2, 16, true>"}
# the other BASELINE configs, run as short legs behind the timed region of the default workload (fresh child processes, never `value`)
OTHER_WORKLOADS = {"sdr1080_24to60": 24, "sdr1080_64pairs": 24, "hdr2160_nb10_blur32": 8, "hdr1080_24to120": 12, "sdr2160_24to60": 8,
                   "sdr360_24to60": 64}   # name: steps (about 1 s timed each); 360p = the size of BASELINE config 1 (the reference's CPU-runnable case)
# Content classes (hopperrender_amd/synth.py ContentScene; SURVEY.md 8(d) "extra cases"): the reference's cost does not depend on the pixels,
# this build's does (staged-warp window fit, SAD reuse, gather coherence), so the line says what content it ran on and what the others cost
CONTENT_LEGS = {"hdr2160_24to120": 6, "sdr1080_24to60": 16}                                   # workload: steps of each scene's leg


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


def device_sysfs(dev_index):
    """sysfs directory of the HIP device (via its PCI address), or None."""
    import glob
    import torch
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        d = "/sys/bus/pci/devices/" + bdf
        if os.path.isdir(d):
            return d
    except Exception:
        pass
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
    return cards[dev_index] if dev_index < len(cards) else None


def device_sample(sysfs):
    """Clock levels and power of the device right now (cheap sysfs reads, no child process): what a reader needs to tell a slow box from a
    regression.  pp_dpm_*: the level marked '*' is the current one."""
    import glob
    if not sysfs:
        return None
    out = {}
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
        txt = _read(os.path.join(sysfs, name))
        if txt:
            cur = [l for l in txt.splitlines() if l.rstrip().endswith("*")]
            try:
                out[key] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit())) if cur else None
            except Exception:
                out[key] = None
    for hw in glob.glob(os.path.join(sysfs, "hwmon", "hwmon*")):
        for key, name in (("power_w", "power1_average"), ("power_w", "power1_input"), ("power_cap_w", "power1_cap"), ("temp_c", "temp1_input")):
            v = _read(os.path.join(hw, name))
            if v and v.lstrip("-").isdigit() and key not in out:
                out[key] = round(int(v) / (1e6 if key.startswith("power") else 1e3), 1)
    return out


def hbm_probe(dev_index):
    """What THIS box's memory system sustains for plain streams, on the idle device before anything else runs: a 2 GiB fill and a 2 GiB ->
    2 GiB copy of torch tensors (best of 5, HIP events).  The boxes of the pool run this pipeline up to 9 % apart at the same shader clock
    (device.shader_clock_mhz_under_load) -- and the same few per cent apart on these two lines, so a reader can normalise `value`."""
    import torch
    try:
        n = 1 << 29
        a = torch.empty(n, dtype=torch.float32, device=f"cuda:{dev_index}")
        b = torch.empty_like(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.fill_(1.0); b.copy_(a); torch.cuda.synchronize(dev_index)
        fill = copy = 0.0
        for _ in range(5):
            e0.record(); a.fill_(2.0); e1.record(); e1.synchronize()
            fill = max(fill, 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            e0.record(); b.copy_(a); e1.record(); e1.synchronize()
            copy = max(copy, 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        out = {"fill_GBps": round(fill, 1), "copy_GBps_read_plus_write": round(copy, 1), "bytes": 4 * n,
               "note": "torch fill_ / copy_ of 2 GiB tensors; streaming_copy_GBps = the library's own 16-bytes-per-lane non-temporal copy (hf_hbm_copy_probe), read + write"}
        try:
            import ctypes as C
            from hopperrender_amd import capi
            g = C.c_double(0.0)
            if capi.load().hf_hbm_copy_probe(dev_index, 4 * n, 5, C.byref(g)) == 0:
                out["streaming_copy_GBps"] = round(g.value, 1)
        except Exception as e:
            out["streaming_copy_error"] = repr(e)[:120]
        return out
    except Exception as e:
        return {"error": repr(e)[:200]}


def device_block(dev_index, sysfs, start, end):
    """The `device` object of the line: which GPU this was, and its clocks / power when the timed region started and ended.  The boxes of the
    pool differ by up to 9 % on this pipeline: compare frames/s across lines only with these fields side by side."""
    import hashlib
    import torch
    d = {}
    try:
        p = torch.cuda.get_device_properties(dev_index)
        d.update({"name": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
                  "memory_gib": round(p.total_memory / 2 ** 30, 1), "clock_rate_khz_max": getattr(p, "clock_rate", None),
                  "pci": "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)})
        uid = getattr(p, "uuid", None)
        d["uuid"] = str(uid) if uid is not None else None
    except Exception as e:
        d["error"] = repr(e)[:200]
    uniq = _read(os.path.join(sysfs, "unique_id")) if sysfs else None
    d["unique_id"] = uniq
    # a stable box id: the GPU's own serial (unique_id) where the driver exposes it, else host name + PCI address
    import socket
    d["box_id"] = hashlib.sha256((uniq or (socket.gethostname() + str(d.get("pci")))).encode()).hexdigest()[:12]
    d["hostname"] = socket.gethostname()
    d["vbios"] = _read(os.path.join(sysfs, "vbios_version")) if sysfs else None
    d["at_start_of_timed_region"] = start
    d["at_end_of_timed_region"] = end
    d["note"] = ("sclk / mclk = the pp_dpm level in force when sampled: right before the timed region (behind the warm-up's synchronisation) and right behind "
                 "its final synchronisation, never inside it; power_cap_w = "
                 "hwmon power1_cap; the chip lowers its clock under load (MI355X_MICROARCH.md 'DVFS give-back'), and devices of the pool differ")
    return d


def warp_symbol(hdr, H=2160, W=3840):
    """Name of the dominant kernel as `rocprofv3 --kernel-trace` prints it, taken from the symbol table of the library this process
    loaded (nm -C), so that the bench line can never name a kernel the binary does not contain."""
    from hopperrender_amd import capi
    prefix = WARP_SYMBOL_PREFIX[(1 if hdr else 0, H * W > 1920 * 1088)]
    if not hdr and H * W <= 640 * 360:          # 8-bit frames at rs = 1: two-pixel flow cells, four per 8-byte thread
        prefix = "warp_fast_kernel<unsigned char, 2, 2, 2, 8, true>"
    try:
        return capi.kernel_symbol(prefix)
    except Exception as e:   # (no `nm` on the box: the measurement must not die for a label)
        return prefix + " ...> (symbol table not readable: %s)" % type(e).__name__


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--periods-per-step", type=int, default=64, help="consecutive source periods per pair stream in one step")
    ap.add_argument("--workload", default="hdr2160_24to120", choices=sorted(WORKLOADS))
    ap.add_argument("--radius", type=int, default=16)
    ap.add_argument("--neighbor", type=int, default=6)
    ap.add_argument("--blur-radius", type=int, default=4)
    ap.add_argument("--streams", type=int, default=0, help="independent frame-pair streams per GPU (0: the workload's operating point)")
    ap.add_argument("--batch", type=int, default=0,
                    help="pair streams per hf_batch: their phase planes, refinement chains and period warps run as one set of launches on "
                         "one HIP stream; streams/batch batches run side by side (0: the workload's operating point; 1: no batching)")
    ap.add_argument("--pool", type=int, default=6, help="distinct synthetic source frames resident in HBM, per pair stream")
    ap.add_argument("--pool-order", default="pingpong", choices=["pingpong", "wrap"],
                    help="how a stream walks its pool: pingpong (0 .. n-1 .. 0: every pair = consecutive frames of the scene) or wrap (p mod n: rounds 1-5, "
                         "a jump of n - 1 frames every n-th pair)")
    ap.add_argument("--py-period-calls", action="store_true", help="A-B: three C calls per batch and period, marshalled inside the timed loop (round 2)")
    ap.add_argument("--member-warps", action="store_true", help="A-B: one fused warp launch and one phase-plane launch per member instead of per batch")
    ap.add_argument("--dual-stream-contexts", action="store_true", help="A-B: HF_FLAG_DUAL_STREAM members (warps overlap the context's own chain)")
    ap.add_argument("--no-fused-warp", action="store_true", help="A-B: one warp launch per output frame instead of one per source period")
    ap.add_argument("--eager-planes", action="store_true", help="A-B: HF_FLAG_BATCH_EAGER_PLANES, every phase plane built by the stand-alone kernel when its frame arrives")
    ap.add_argument("--no-sad-reuse", action="store_true", help="A-B: HF_FLAG_NO_SAD_REUSE, every refinement step recomputes its candidate SADs as the reference does")
    ap.add_argument("--no-lazy-argmin", action="store_true", help="A-B: HF_FLAG_NO_LAZY_ARGMIN, 6 more (tiny) launches per flow chain")
    ap.add_argument("--timing-events", action="store_true",
                    help="keep the reference's per-call timing events (m_ofcCalcTime, m_warpCalcTime); default off in the bench")
    ap.add_argument("--copy-in", action="store_true",
                    help="updateFrame copies the device-resident source frame into the ring (default: zero-copy reference)")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="bracket every n-th warp launch / flow chain with HIP events (event records perturb back-to-back launches)")
    ap.add_argument("--diagnose", default="", choices=["", "no-flow", "no-warp"],
                    help="NOT a benchmark: drop the flow chain or the warps from the step to see what the other part costs")
    ap.add_argument("--pg-first", action="store_true",
                    help="A-B: create the process group (a single-rank RCCL communicator when not launched by torchrun) BEFORE the "
                         "batch streams, to see what the communicator's streams do to the hardware-queue layout")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not sample the shader clock in the middle of the timed region")
    ap.add_argument("--timeline-out", default="", help="DIAGNOSTIC: record the start / stop of every dispatch of every batch stream for --timeline-steps "
                    "steps in the middle of the timed region (hf_batch_timeline_*: no profiler) plus a stand-alone leg of one batch, and write "
                    "the raw records + tools/timeline_report.py's analysis to this JSON file")
    ap.add_argument("--timeline-steps", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--no-host-io", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short legs of the other BASELINE configs behind the default workload")
    ap.add_argument("--cpu-sample-pairs", type=int, default=8)
    ap.add_argument("--scene", default="bench", choices=["bench", "static", "pan64", "chaotic", "cut"],
                    help="content class of the synthetic source frames (hopperrender_amd/synth.py ContentScene): bench = global (+7, -3) px per frame + "
                         "12 rectangles up to +-48 px; static; pan64 = pure 64 px pan; chaotic = every 16x16 block its own motion up to +-96 px on "
                         "full-range noise; cut = every pair a hard cut")
    ap.add_argument("--content-counters", action="store_true", help="behind the timed region: device-side counters of a few periods (share of staged / "
                    "global / generic warp workgroups, share of windows that reused their SADs per level; include/hopperflow_diag.h)")
    ap.add_argument("--no-content-legs", action="store_true", help="skip the short legs of the other content classes behind the default workload")
    return ap.parse_args()


def cpu_baseline(hdr, H, W, target, radius, neighbor, frames, n_pairs):
    """The oracle port (single thread) on a bounded sample of the same workload."""
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule
    from oracle import oracle
    g = oracle.make_geom(hdr, H, W)
    plan = BlendSchedule(SOURCE_24, target).plan(n_pairs + 1)[1:]
    t0 = time.perf_counter()
    outs = 0
    flow_s = 0.0
    for i in range(n_pairs):
        f0, f1, f2 = frames[i % len(frames)], frames[(i + 1) % len(frames)], frames[(i + 2) % len(frames)]
        tf = time.perf_counter()
        _, blur, _, _ = oracle.calculate_optical_flow(f1, f2, g, radius, 0, 8, neighbor, 4)
        flow_s += time.perf_counter() - tf
        for t in plan[i]:
            oracle.warp_frames(f0, f1, blur, g, t, 2)
            outs += 1
    dt = time.perf_counter() - t0
    return {"value": outs / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "ms_per_flow_calc": 1e3 * flow_s / n_pairs,
            "sample": f"{n_pairs} source pairs ({outs} output frames) of the same workload, oracle/hf_oracle.c, 1 thread, {dt:.1f} s"}


def cpu_baseline_all_cores(hdr, H, W, target, radius, neighbor, frames):
    """The same oracle port with one source pair per host thread (pairs are independent units, SURVEY.md section 8(e);
    the C calls release the GIL), on EVERY core the process may run on."""
    from concurrent.futures import ThreadPoolExecutor
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule
    from oracle import oracle
    g = oracle.make_geom(hdr, H, W)
    threads = max(1, len(os.sched_getaffinity(0)))
    n_pairs = threads
    plan = BlendSchedule(SOURCE_24, target).plan(n_pairs + 1)[1:]

    def one(i):
        f0, f1, f2 = frames[i % len(frames)], frames[(i + 1) % len(frames)], frames[(i + 2) % len(frames)]
        _, blur, _, _ = oracle.calculate_optical_flow(f1, f2, g, radius, 0, 8, neighbor, 4)
        for t in plan[i]:
            oracle.warp_frames(f0, f1, blur, g, t, 2)
        return len(plan[i])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        outs = sum(ex.map(one, range(n_pairs)))
    dt = time.perf_counter() - t0
    return {"value": outs / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n_pairs} source pairs ({outs} output frames), one pair per thread, {threads} threads = all host cores, {dt:.1f} s"}


def reference_opencl(hdr, H, W, target, radius, neighbor, frames):
    """The reference's own OpenCL path on this GPU (child process; outside the timed region)."""
    from hopperrender_amd.protocol import SOURCE_24
    from oracle import oracle
    if not oracle.ref_available():
        return None
    s = oracle.RefSession(hdr, H, W, 0, 0, 8, neighbor, 0.0, 255.0, 270)
    s.radius(radius)
    for f in frames[:4]:
        s.update(f)
    s.calc(); s.calc()
    s.time_calc(200); s.time_warp(400, 0.3996, 2)
    js, _ = s.run(timeout=300)
    calc_ms, warp_ms = js[0]["time_calc_ms"], js[1]["time_warp_ms"]
    k = SOURCE_24 / target
    return {"ms_per_flow_calc": calc_ms, "ms_per_warp": warp_ms, "frames_per_s": 1e3 * k / (calc_ms + k * warp_ms),
            "note": "reference host code + OpenCL kernels (oracle/_ref) on the same MI355X, device time of "
                    "back-to-back calculateOpticalFlow / warpFrames calls, no host transfers"}


def host_io_block(hdr, H, W, target, n_periods=24, device=0, async_only=False):
    """PCIe-inclusive rates of ONE context fed from and read back into host memory (never the bench `value`): the
    reference's blocking protocol with pageable and with pinned buffers, and the asynchronous variant
    (hf_update_frame_async / hf_download_frame_async: pinned buffers, H2D and D2H on side streams).  Measured by
    tools/host_io_rate.py in a CHILD process: this process still holds its pair streams, and HIP deals all streams of a
    process onto a handful of hardware queues (the four streams of an asynchronous context would share queues with them:
    1080p SDR 3.0 k instead of 8.6-9.0 k frames/s)."""
    import subprocess
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_io_rate.py"), "--hdr", str(hdr), "--H", str(H), "--W", str(W),
                        "--target", str(target), "--n", str(n_periods), "--device", str(device)] + (["--async-only"] if async_only else []),
                       capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    res["note"] = (f"one context, {n_periods} source periods, frames enter and leave through host memory (PCIe Gen5 x16), every output "
                   "frame returned to the host; child process; never the bench `value`")
    return res


def other_workloads(a, budget_s=200.0):
    """BASELINE configs 2, 4 and 5 as ~1 s legs of this same script in fresh child processes (their own HIP runtime and hardware
    queues), AFTER the timed region of the default workload: what the driver's one bench line would otherwise never show.  Reported
    next to `value`, never part of it.  ONE shared deadline (`budget_s`) bounds what the legs add to the run; a leg that fails or does
    not fit is named in `failed` (and on stderr) -- a regression of configs 2, 4 or 5 must not hide inside an entry."""
    import subprocess
    res, failed = {}, []
    deadline = time.monotonic() + budget_s
    for name, steps in OTHER_WORKLOADS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", "2", "--radius", str(a.radius),
               "--no-cpu-baseline", "--no-reference", "--no-host-io", "--no-other-workloads", "--no-content-legs"]
        left = deadline - time.monotonic()
        try:
            if left < 20.0:
                raise TimeoutError(f"skipped: {left:.0f} s of the legs' shared {budget_s:.0f} s budget left")
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
            if r.returncode != 0:
                raise RuntimeError(f"exit status {r.returncode}: {r.stderr[-300:]}")
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            roof = d["roofline"]
            res[name] = {"value": d["value"], "unit": d["unit"], "frac": roof["frac"], "frac_algorithmic": roof["frac_algorithmic"],
                         "frac_compulsory": roof.get("frac_compulsory"),
                         "ms_per_flow_calc": d["ms_per_flow_calc"], "timed_region_s": d["timed_region_s"], "steps": d["steps"],
                         "pair_streams": d["config"]["pair_streams_total"], "flow_batch": d["config"]["flow_batch"],
                         "kernel": roof["kernel"], "bytes_per_output_frame": (roof.get("traffic_pipeline") or {}).get("hbm_bytes_per_output_frame")}   # (None: no PMC record of this workload)
        except Exception as e:
            res[name] = {"error": repr(e)[:300]}
            failed.append(name)
            print(f"bench.py: other_workloads leg {name} failed: {e!r}"[:500], file=sys.stderr)
    res["failed"] = failed
    return res


def content_legs(a, budget_s=200.0):
    """The other content classes as ~1 s legs of this same script (fresh child processes behind the timed region of the default workload), at
    2160p HDR and 1080p SDR: frames/s, chain us per pair (in the pipeline and alone), and the device-side counters of what the kernels decided.
    Reported next to `value`, never part of it."""
    import subprocess
    res, failed = {}, []
    deadline = time.monotonic() + budget_s
    for wl, steps in CONTENT_LEGS.items():
        res[wl] = {}
        for scene in ("bench", "bench_wrap6", "static", "pan64", "chaotic", "cut"):
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--scene", scene.split("_")[0], "--steps", str(steps), "--warmup", "2", "--radius", str(a.radius),
                   "--content-counters", "--no-cpu-baseline", "--no-reference", "--no-host-io", "--no-other-workloads", "--no-content-legs"]
            if scene == "bench_wrap6":
                cmd += ["--pool-order", "wrap", "--pool", "6"]      # the frame sequence of the bench lines of rounds 1-5
            left = deadline - time.monotonic()
            try:
                if left < 15.0:
                    raise TimeoutError(f"skipped: {left:.0f} s of the legs' shared {budget_s:.0f} s budget left")
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
                if r.returncode != 0:
                    raise RuntimeError(f"exit status {r.returncode}: {r.stderr[-300:]}")
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                res[wl][scene] = {"value": d["value"], "unit": d["unit"], "us_per_flow_calc_in_pipeline": round(1e3 * (d["ms_per_flow_calc"] or 0), 2),
                                  "us_per_flow_calc_alone": round(1e3 * (d["ms_per_flow_calc_isolated"] or 0), 2), "steps": d["steps"],
                                  "counters": d.get("content_counters")}
            except Exception as e:
                res[wl][scene] = {"error": repr(e)[:300]}
                failed.append(f"{wl}/{scene}")
                print(f"bench.py: content leg {wl}/{scene} failed: {e!r}"[:500], file=sys.stderr)
    res["failed"] = failed
    res["note"] = ("bench = this line's scene; bench_wrap6 = the same scene in the frame order of rounds 1-5 (pool of 6 walked modulo: a 5-frame jump every 6th pair); per content class: whole-job frames/s of a short run, flow chain device time per pair, and counters of 4 periods of one batch: warp_staged_share = "
                   "workgroups of the fused period warp that staged their window in LDS (the rest: interior global path / generic body), reuse_share[window][axis] = "
                   "windows of full tiles that summed their blocks' SAD vectors instead of gathering (csrc/hf_flow.hip)")
    return res


def main():
    a = parse_args()
    result_line = ""
    import numpy as np
    import torch  # first: one HIP runtime for the whole process
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HF_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices; the timing reductions run on CPU tensors).  Default: nccl (= RCCL).
    backend = os.environ.get("HF_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)

    def init_dist():
        # Called AFTER the pair streams exist: HIP deals streams to its few hardware queues in creation order, and the
        # communicator's own streams must not push two batch streams onto one queue (DESIGN.md "Hardware queues").
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend)
    pg_done = False
    if a.pg_first:
        if world > 1:
            init_dist()
        else:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
            dist.all_reduce(torch.zeros(1, device="cuda"))       # forces the communicator (and its streams) into existence
        pg_done = True
    n_gpus = world
    if a.gpus != n_gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import __graft_entry__
    __graft_entry__.build(quiet=True)
    hbm_idle = hbm_probe(dev_index) if rank == 0 else None     # before any stream of the workload exists
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, BlendSchedule

    hdr, H, W, target, desc = WORKLOADS[a.workload]
    dev = dev_index
    for k, v in WORKLOAD_PARAMS.get(a.workload, {}).items():
        setattr(a, k, v)
    if a.workload in TOTAL_PAIRS:
        per_rank = max(1, TOTAL_PAIRS[a.workload] // world)
        op_streams, op_batch = per_rank, min(per_rank, 16)     # (64 pairs on one GPU: 4 batch streams of 16 -- 126 k against 121.5 k frames/s as 2 x 32)
    else:
        op_streams, op_batch = OPERATING_POINT[a.workload]
    if a.streams <= 0:
        a.streams = op_streams
    if a.batch <= 0:
        a.batch = min(op_batch, a.streams)
    if a.streams % a.batch:
        raise SystemExit("--streams must be a multiple of --batch")

    # ---- synthetic source frames, resident in HBM before the timed region ----
    scene = synth.ContentScene(a.scene, H, W, bool(hdr), seed=1234 + rank)     # (bench: the synth.Scene of rounds 1-5, frame for frame)
    host_frames = [scene.frame(k) for k in range(a.pool)]
    # every pair stream gets its OWN device copies (no artificial cache sharing between streams); the
    # streams start at different frames of the sequence
    pools = []
    for s in range(a.streams):
        bufs = []
        for f in host_frames:
            b = DeviceBuffer(f.nbytes, dev)
            b.upload(f)
            bufs.append(b)
        pools.append(bufs)

    def pool_index(p):
        """Frame of the pool a stream shows at period p.  pingpong (default): 0 1 .. n-1 n-2 .. 1 0 1 ..: EVERY pair is a pair of consecutive frames of
        the scene, forwards or backwards, as SURVEY.md 8(d) defines the synthetic pairs.  wrap (rounds 1-5): p mod n -- every n-th pair (5 -> 0) jumps
        n - 1 frames back: a hard cut every sixth period that no earlier bench line mentioned (content leg "bench_wrap6" keeps it measured)."""
        n = a.pool
        if a.pool_order == "wrap" or n < 3:
            return p % n
        q = p % (2 * (n - 1))
        return q if q < n else 2 * (n - 1) - q

    flags = capi.HF_FLAG_ASYNC | (0 if a.no_profile else capi.HF_FLAG_PROFILE)
    if not a.timing_events:
        flags |= capi.HF_FLAG_NO_TIMING   # the m_ofcCalcTime / m_warpCalcTime events are barrier packets between the kernels
    if a.no_fused_warp:
        flags |= capi.HF_FLAG_NO_FUSED_WARP
    if a.dual_stream_contexts:
        flags |= capi.HF_FLAG_DUAL_STREAM
    if a.no_lazy_argmin:
        flags |= capi.HF_FLAG_NO_LAZY_ARGMIN
    if a.no_sad_reuse:
        flags |= capi.HF_FLAG_NO_SAD_REUSE
    if a.eager_planes:
        flags |= capi.HF_FLAG_BATCH_EAGER_PLANES
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    calcs, outbufs, plans = [], [], []
    P = a.periods_per_step
    BURST = 8                                    # periods of the host-cost probe behind the timed region
    total_periods = (a.warmup + a.steps) * P + BURST
    max_out = 6 if target == 83333 else 3
    schedule = BlendSchedule(SOURCE_24, target).plan(total_periods + 3)[3:]
    for s in range(a.streams):
        c = cls(H, W, 0, 0, 8, a.neighbor, 0.0, 255.0, 270, device_index=dev, search_radius=a.radius,
                blur_radius=a.blur_radius, flags=flags)
        calcs.append(c)
        if not a.no_profile:
            c.setProfileInterval(a.profile_every, max(1, a.profile_every // 4))
        outbufs.append([DeviceBuffer(c.output_frame_bytes, dev) for _ in range(max_out)])
        plans.append(schedule)
        for k in range(3):  # prime the 3-frame ring and the previous-flow slot (m_frameCount >= 3)
            c.updateFrameDeviceRef(pools[s][pool_index(s + k)].ptr)
        c.calculateOpticalFlow()
        c.sync()

    out_ptrs = [[b.ptr for b in bufs] for bufs in outbufs]
    batches = [FlowBatch(calcs[k:k + a.batch]) for k in range(0, a.streams, a.batch)] if a.batch > 1 else []

    def src_ptr(s, i):
        return pools[s][pool_index(s + 3 + i)].ptr

    # The schedule of a throughput driver is known ahead of time: the arguments of every hf_batch_run_period call (ONE native
    # call per batch and source period -- include/hopperflow.h) are marshalled before the timed region, as a C host's would be.
    one_call = bool(batches) and not (a.member_warps or a.copy_in or a.diagnose or a.py_period_calls)
    deferred_planes = one_call and batches[0].defersPlanes()
    prepared, prepared_frames = {}, {}
    if one_call:
        for i in range(total_periods):
            for bi, b in enumerate(batches):
                lo, hi = bi * a.batch, (bi + 1) * a.batch
                prepared[(i, bi)] = b.preparePeriod([src_ptr(s, i) for s in range(lo, hi)], [plans[s][i] for s in range(lo, hi)], out_ptrs[lo:hi], 2)
                prepared_frames[(i, bi)] = sum(len(plans[s][i]) for s in range(lo, hi))

    def run_period_batched(i):
        """Per batch: the new source frame of every member pair (one phase-plane launch), ONE batched flow calculation,
        then every member's outputs of the period in ONE fused warp launch -- all on the batch's stream."""
        n = 0
        if one_call:
            for bi, b in enumerate(batches):
                b.runPeriod(prepared[(i, bi)])
                n += prepared_frames[(i, bi)]
            return n
        for bi, b in enumerate(batches):
            lo, hi = bi * a.batch, (bi + 1) * a.batch
            if a.member_warps or a.copy_in:
                for s in range(lo, hi):
                    (calcs[s].updateFrameDevice if a.copy_in else calcs[s].updateFrameDeviceRef)(src_ptr(s, i))
            else:
                b.updateFramesDeviceRef([src_ptr(s, i) for s in range(lo, hi)])
            if a.diagnose != "no-flow":
                b.calculateOpticalFlow()
            if a.diagnose != "no-warp":
                if a.member_warps:
                    for s in range(lo, hi):
                        calcs[s].interpolateOnly(plans[s][i], out_ptrs[s], 2)
                else:
                    b.interpolatePeriod([plans[s][i] for s in range(lo, hi)], out_ptrs[lo:hi], 2)
            n += sum(len(plans[s][i]) for s in range(lo, hi))
        return n

    def run_period(i):
        if batches:
            return run_period_batched(i)
        n = 0
        for s, c in enumerate(calcs):
            ts = plans[s][i]
            if a.diagnose == "no-flow":
                c.updateFrameDeviceRef(src_ptr(s, i))
                c.interpolateOnly(ts, out_ptrs[s], 2)
            elif a.diagnose == "no-warp":
                c.interpolatePeriod(src_ptr(s, i), [], [], 2)
            elif a.copy_in:
                c.updateFrameDevice(src_ptr(s, i))
                c.interpolatePeriod(0, ts, out_ptrs[s], 2)
            else:
                c.interpolatePeriod(src_ptr(s, i), ts, out_ptrs[s], 2)
            n += len(ts)
        return n

    def run_step(k):
        n = 0
        for i in range(k * P, (k + 1) * P):
            n += run_period(i)
        return n

    def sync_all():
        for c in calcs:
            c.sync()
        torch.cuda.synchronize()

    timeline_on = bool(a.timeline_out) and bool(batches) and rank == 0
    if timeline_on:
        # armed before the warm-up (enable synchronises the batch's stream), recording from the middle of the timed region
        tl_periods = max(2, int(round(a.timeline_steps * P)))
        tl_skip = (a.warmup + a.steps // 2) * P
        for b in batches:
            b.timelineEnable(tl_periods * 16 + 32, tl_skip)
    if not pg_done:
        init_dist()
    rank_devices = [dev_index]
    if world > 1:     # which device every rank opened (rank r -> r % device_count): part of the line, checked by the multi-rank tests
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, dev_index)
    clock_under_load = clock_probe_call_ms = None
    import ctypes as C
    for k in range(a.warmup):
        run_step(k)
        if rank == 0 and k == a.warmup - 1 and not a.no_clock_probe:
            # The shader clock the device sustains under THIS load: a one-wave probe on a stream of its own, 3 ms (the clock moves with the load from millisecond to millisecond), issued behind the last
            # warm-up step while the GPU is still working through it.  NOT inside the timed region: its normal-priority stream only gets
            # through when the highest-priority batch streams leave a gap (measured: the call returns after 0.1-0.7 s), and the issuing
            # loop must not stand still that long while it is being timed.
            mhz = C.c_double(0.0)
            tp = time.perf_counter()
            if capi.load().hf_clock_probe(dev_index, 3000, C.byref(mhz)) == 0:
                clock_under_load = round(mhz.value, 1)
            clock_probe_call_ms = round(1e3 * (time.perf_counter() - tp), 3)
    sync_all()
    for c in calcs:
        if not a.no_profile:
            c.resetProfile()

    sysfs = device_sysfs(dev_index) if rank == 0 else None
    # clock levels / board power right before and right behind the timed region, never inside it: the sysfs reads are SMU queries of up to
    # milliseconds, and only rank 0 makes them (what the device sustains UNDER the load is the clock probe's figure, taken behind the warm-up)
    dev_mid = [device_sample(sysfs)] if rank == 0 else []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames_out = 0
    for k in range(a.warmup, a.warmup + a.steps):
        frames_out += run_step(k)
    host_issue_wall_s = time.perf_counter() - t0   # the host is done issuing; the GPU may still be busy.  NOT the host's cost:
                                                   # once the hardware queues are full every further call blocks until the GPU
                                                   # has retired a packet, so this wall time tracks the GPU's
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        dev_mid.append(device_sample(sysfs))
    # What the host loop itself costs: BURST periods issued into EMPTY queues (nothing blocks), per native call
    tb = time.perf_counter()
    for i in range((a.warmup + a.steps) * P, (a.warmup + a.steps) * P + BURST):
        run_period(i)
    host_call_s = (time.perf_counter() - tb) / BURST     # per source period of all streams of this rank
    sync_all()
    host_enqueue_s = host_call_s * P * a.steps

    content_counters = None
    if a.content_counters and rank == 0 and batches:
        # what the kernels decided on THIS content: counters of 4 periods of the first batch (device-side atomics, include/hopperflow_diag.h);
        # behind the timed region, because the counting waves issue atomics
        lead = calcs[0]
        lead.countersEnable(True)
        b0 = batches[0]
        base = (a.warmup + a.steps) * P + BURST
        sched_c = BlendSchedule(SOURCE_24, target).plan(base + 8)[base:]
        for i in range(4):
            b0.runPeriod(b0.preparePeriod([src_ptr(s, base + i) for s in range(a.batch)], [sched_c[i] for _ in range(a.batch)], out_ptrs[:a.batch], 2))
        b0.sync()
        cc = lead.counters()
        lead.countersEnable(False)
        wg = cc["warp_workgroups"]
        tot_wg = max(sum(wg.values()), 1)
        st_lead = lead.stats()
        content_counters = {"sad_tables": st_lead["sad_tables"], "still_share_of_32_windows": round(st_lead["still_share"], 4), "warp_workgroups": wg, "warp_staged_share": round(wg["staged"] / tot_wg, 4) if sum(wg.values()) else None,
                            "reuse_share": {str(w): {ax: round(v[1] / max(v[0], 1), 4) for ax, v in lv.items()} for w, lv in sorted(cc["levels"].items(), reverse=True)},
                            "windows_counted": {str(w): lv["X"][0] for w, lv in sorted(cc["levels"].items(), reverse=True)}, "periods": 4, "members": a.batch}

    red_dev = "cuda" if backend == "nccl" else "cpu"
    tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    ft = torch.tensor([float(frames_out)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
    elapsed_max, frames_total = float(tt.item()), float(ft.item())

    prof = {"warp_launches": 0, "warp_ms": 0.0, "flow_chains": 0, "flow_ms": 0.0, "warp_frames": 0}
    if not a.no_profile:
        for c in calcs:
            p = c.profile()
            for k in prof:
                prof[k] += p[k]

    timeline = None
    if timeline_on:
        streams = [b.timelineRead() for b in batches]
        for b in batches:
            b.timelineEnable(0)
        # the same launches with ONE batch stream running and the others idle: stand-alone durations
        b0 = batches[0]
        b0.timelineEnable(12 * 16 + 32, 2)
        base = (a.warmup + a.steps) * P + BURST
        sched_alone = BlendSchedule(SOURCE_24, target).plan(base + 20)[base:]
        for i in range(14):
            b0.runPeriod(b0.preparePeriod([src_ptr(s, base + i) for s in range(a.batch)], [sched_alone[i] for _ in range(a.batch)], out_ptrs[:a.batch], 2))
        b0.sync()
        alone = b0.timelineRead()
        b0.timelineEnable(0)
        timeline = {"streams": streams, "alone": alone, "periods_per_step": P}

    # Context for the roofline figure, OUTSIDE the timed region: the dominant kernel alone on the GPU -- one batch of a.batch members,
    # i.e. exactly the launch the pipeline issues (same kernel, same grid), one fused period launch at a time, source frames
    # rotating so that they come from HBM -- and the flow chain of one pair alone.
    isolated = None
    if rank == 0 and not a.no_profile:
        for b in batches:
            b.close()
        batches = []
        nb = min(max(a.batch, 1), len(calcs))
        c = calcs[0]
        for x in calcs[:nb]:
            x.setProfileInterval(1, 1)
            x.resetProfile()
        for i in range(12):                            # the chain of ONE pair alone, each behind a host synchronisation (an idle device)
            c.updateFrameDeviceRef(pools[0][pool_index(i)].ptr)
            c.calculateOpticalFlow()
            c.sync()
        flow_us_after_sync = 1e3 * c.profile()["flow_ms"] / max(c.profile()["flow_chains"], 1)
        c.resetProfile()
        for i in range(48):                            # ... and back to back: device time of each chain (still a new frame, i.e. a cold plane, every time)
            c.updateFrameDeviceRef(pools[0][pool_index(i)].ptr)
            c.calculateOpticalFlow()
        c.sync()
        flow_us = 1e3 * c.profile()["flow_ms"] / max(c.profile()["flow_chains"], 1)
        c.resetProfile()
        for i in range(48):                            # ... and on ONE frame pair again and again (planes cache-warm): the way reference_opencl.ms_per_flow_calc is taken
            c.calculateOpticalFlow()
        c.sync()
        flow_us_warm = 1e3 * c.profile()["flow_ms"] / max(c.profile()["flow_chains"], 1)
        small = FlowBatch(calcs[:nb]) if nb > 1 else None
        for x in calcs[:nb]:
            x.resetProfile()
        for i in range(24):
            ts = [plans[s][i % len(plans[s])] for s in range(nb)]
            if small:
                # one whole period per call, as in the pipeline, one batch stream only: its launches run one after the other, so
                # the fused warp launch (with the plane-building workgroups of a deferred-plane batch) has the GPU to itself; the
                # profile spans time the dispatch, not the call
                small.runPeriod(small.preparePeriod([pools[s][pool_index(i + s)].ptr for s in range(nb)], ts, out_ptrs[:nb], 2))
                small.sync()
            else:
                c.updateFrameDeviceRef(pools[0][pool_index(i)].ptr); c.calculateOpticalFlow(); c.sync()
                c.interpolateOnly(ts[0], out_ptrs[0], 2); c.sync()
        p = c.profile()                                # the leader's profile carries the batch's warp launches
        if small:
            small.close()
        isolated = {"warp_us": 1e3 * p["warp_ms"] / max(p["warp_launches"], 1) / nb, "fpl": p["warp_frames"] / max(p["warp_launches"], 1) / nb,
                    "launch_us": 1e3 * p["warp_ms"] / max(p["warp_launches"], 1), "flow_chain_us": flow_us, "flow_chain_after_sync_us": flow_us_after_sync, "flow_chain_warm_us": flow_us_warm, "members": nb}

    # Host-I/O leg at N > 1 (the expected scaling limit is host memcpy / the PCIe root complex): EVERY rank feeds
    # one context from pinned host memory and reads every output frame back, all ranks at the same time, each in a child process
    # on its own GPU; rank 0 reports the aggregate.  Never part of `value`.
    host_io_ranks = None
    if world > 1 and not a.no_host_io:
        dist.barrier()
        try:
            mine = host_io_block(hdr, H, W, target, n_periods=24, device=dev_index, async_only=True)
        except Exception as e:
            mine = {"error": repr(e)}
        host_io_ranks = [None] * world
        dist.all_gather_object(host_io_ranks, mine)

    if rank == 0:
        st = calcs[0].stats()
        N = st["low_width"] * st["low_height"]
        F = st["output_frame_bytes"]
        b_out = 3 * F + 4 * N  # read 2 source frames + write 1 + blurred flow once
        value = frames_total / elapsed_max
        pipeline_gbs = value / n_gpus * b_out / 1e9              # per GPU: the roofline is a per-device bound
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                # configs 4 / 5 run the same kernels on the same frame geometry as configs 2 / 3: they are priced with those PMC records
                traffic_key = {"sdr1080_64pairs": "sdr1080_24to60", "hdr2160_nb10_blur32": "hdr2160_24to120"}.get(a.workload, a.workload)
                traffic = json.load(open(tpath)).get(traffic_key, {})
            except Exception:
                traffic = None
        # PHYSICAL roofline: HBM bytes the whole pipeline moves per output frame (warp + phase planes + chain, rocprofv3 PMC passes
        # over this very command at this operating point, generated into profiles/roofline_traffic.json by tools/pmc_traffic.py
        # --pipeline) x frames/s.  The SURVEY 8(d) figure (3F + 4N "algorithmic" bytes per output frame, which a fused period
        # launch legitimately does not move: it reads the two source frames once for all outputs) is kept as frac_algorithmic.
        pipe = (traffic or {}).get("pipeline") or {}
        bytes_per_frame = pipe.get("hbm_bytes_per_output_frame")
        if bytes_per_frame:
            basis = "measured: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the pipeline of workload %s at %s (profiles/roofline_traffic.json)" % (traffic_key, pipe.get("operating_point"))
        else:   # no PMC record for this workload: the compulsory traffic of the launches (what they cannot avoid moving)
            k_out = (SOURCE_24 / target)
            pp_bytes = st["phase_plane_bytes"]
            bytes_per_frame = ((2 + k_out) * F + 4 * N + F + 3 * pp_bytes) / k_out
            basis = "model (no PMC record for this workload): per source period 2F + kF + 4N (warp) + F + plane write + 2 plane reads"
        physical_gbs = value / n_gpus * bytes_per_frame / 1e9
        # The guide's x 2 on FETCH_SIZE is calibrated for WIDE coalesced reads (16 bytes per lane: the warp's window copies, the plane build).
        # The chain's gathers (16-byte segments at dword alignment, 64-byte row pieces) are not such reads: for them x 2 is an upper bound.
        # Both readings are printed: `frac` doubles every kernel's FETCH_SIZE, `frac_narrow_gathers_x1` leaves the chain + blur kernels' as counted.
        per_k_all = pipe.get("per_kernel_bytes_per_pair_and_period") or {}
        k_out_sched = frames_total / max(1.0, float(n_gpus * a.streams * P * a.steps))        # output frames per pair and source period
        narrow = sum(v["read"] / 2 for kname, v in per_k_all.items() if kname.startswith(("flow_", "blur_")))
        bytes_per_frame_x1 = bytes_per_frame - narrow / (pipe.get("output_frames_per_pair_and_period") or k_out_sched) if per_k_all else None
        # COMPULSORY bytes: what no implementation of the path can avoid moving per pair and source period -- the two source
        # frames once (2F), every output frame once (kF), the blurred flow once per output (4N each) and the flow calculation's own
        # B_flow = 6 N bpp + 4 N -- against the same 8 TB/s.
        bpp = 2 if hdr else 1
        b_flow = 6 * N * bpp + 4 * N
        compulsory_pair_period = 2 * F + k_out_sched * F + k_out_sched * 4 * N + b_flow
        compulsory_gbs = value / n_gpus * (compulsory_pair_period / k_out_sched) / 1e9
        roof = {"bound": "hbm", "achieved": round(physical_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(physical_gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_bw_6290": round(physical_gbs / 6290.0, 4),
                "definition": "physical: frames/s per GPU x HBM bytes the pipeline moves per output frame (all kernels)",
                "traffic_record_commit": (traffic or {}).get("git_commit") if pipe else None,
                "traffic_pipeline": {"hbm_bytes_per_output_frame": int(bytes_per_frame), "basis": basis,
                                     "per_kernel_bytes_per_pair_and_period": pipe.get("per_kernel_bytes_per_pair_and_period")},
                "frac_narrow_gathers_x1": round(value / n_gpus * bytes_per_frame_x1 / 1e9 / HBM_PEAK_GBS, 4) if bytes_per_frame_x1 else None,
                "frac_compulsory": round(compulsory_gbs / HBM_PEAK_GBS, 4), "achieved_compulsory": round(compulsory_gbs, 1),
                "compulsory_definition": "frames/s per GPU x (2F + kF + 4N k + B_flow) / k bytes per output frame, k = output frames per pair and source period, "
                                         "B_flow = 6 N bpp + 4 N (SURVEY.md 8(d)): bytes NO implementation can avoid",
                "compulsory_bytes_per_pair_and_period": int(compulsory_pair_period), "moved_over_compulsory": round(bytes_per_frame * k_out_sched / compulsory_pair_period, 3),
                "frac_of_this_box_streaming_copy": (round(physical_gbs / hbm_idle["streaming_copy_GBps"], 4) if hbm_idle and hbm_idle.get("streaming_copy_GBps") else None),
                "frac_of_this_box_note": "achieved (physical GB/s) / what a plain 16-byte-per-lane copy reads + writes per second on the idle device of THIS run "
                                         "(device.hbm_streams_idle_device.streaming_copy_GBps): the pipeline's traffic is 28 % reads and 72 % writes",
                "frac_algorithmic": round(pipeline_gbs / HBM_PEAK_GBS, 4), "achieved_algorithmic": round(pipeline_gbs, 1),
                "algorithmic_definition": "SURVEY.md 8(d): frames/s per GPU x B_out, B_out = 3F + 4N bytes per output frame",
                "algorithmic_bytes_per_unit": b_out,
                "traffic": None, "traffic_note": None,
                "kernel": warp_symbol(hdr, H, W)}
        # HBM bytes of ONE launch of the dominant kernel as the pipeline issues it (a.batch members): PMC pass over the pipeline, else
        # the pass over single-member launches
        per_k = (pipe.get("per_kernel_bytes_per_pair_and_period") or {})
        warp_bytes_member = next((v["read"] + v["write"] for k, v in per_k.items() if k.startswith("warp_")), None)
        if warp_bytes_member:
            roof["traffic"] = int(warp_bytes_member * max(a.batch, 1))
            roof["traffic_note"] = (f"HBM bytes of one launch of the dominant kernel = {max(a.batch, 1)} members' fused periods ({int(warp_bytes_member)} B per member; "
                                    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the pipeline, profiles/roofline_traffic.json generated by tools/pmc_traffic.py)")
        elif traffic and traffic.get("warp_kernel_hbm_bytes_per_launch"):
            roof["traffic"] = traffic["warp_kernel_hbm_bytes_per_launch"]
            roof["traffic_note"] = "HBM bytes of one single-member fused period launch (PMC pass over --streams 1 --batch 1)"
        if prof["warp_launches"]:
            avg_ms = prof["warp_ms"] / prof["warp_launches"]
            fpl = prof["warp_frames"] / prof["warp_launches"]     # output frames per launch (a batch's period is one fused launch)
            roof["kernel_in_pipeline"] = {
                "avg_launch_us": round(avg_ms * 1e3, 2), "output_frames_per_launch": round(fpl, 3),
                "algorithmic_GBps": round(b_out * fpl / (avg_ms * 1e-3) / 1e9, 1), "launches_sampled": prof["warp_launches"],
                "note": "HIP events of the dispatch itself inside the timed region; the other batch streams' kernels share the GPU "
                        "while it runs, so this is neither a kernel roofline nor a pipeline one"}
        if isolated:
            alg = b_out * isolated["fpl"] / (isolated["warp_us"] * 1e-6) / 1e9
            iso = {"avg_launch_us": round(isolated["launch_us"], 2), "us_per_member": round(isolated["warp_us"], 2),
                   "output_frames_per_member": round(isolated["fpl"], 3),
                   "algorithmic_GBps": round(alg, 1), "algorithmic_frac": round(alg / HBM_PEAK_GBS, 4),
                   "members_per_launch": isolated["members"],
                   "note": "the same kernel alone on the GPU (one launch of the fused periods of a whole batch at a time = the launch the pipeline issues, times per member, sources from HBM), after the timed "
                           "region.  'algorithmic' credits 3F + 4N per output frame although the fused launch reads the two source "
                           "frames once for all its outputs; 'real' prices the bytes the PMC counters saw"}
            # bytes one member's period moves through this kernel: from the PMC pass over the pipeline (the kernel the pipeline runs) or,
            # failing that, from the pass over single-member launches
            warp_bytes = warp_bytes_member
            if warp_bytes is None and traffic and abs(traffic.get("units_per_launch", 0) - isolated["fpl"]) < 0.75:
                warp_bytes = traffic.get("warp_kernel_hbm_bytes_per_launch")
            if warp_bytes:
                iso["hbm_bytes_per_member"] = int(warp_bytes)
                real = warp_bytes / (isolated["warp_us"] * 1e-6) / 1e9
                iso["real_GBps"] = round(real, 1)
                iso["real_frac"] = round(real / HBM_PEAK_GBS, 4)
            roof["kernel_isolated"] = iso
        out = {
            **({"DIAGNOSTIC_NOT_A_BENCHMARK": a.diagnose} if a.diagnose else {}),
            "metric": "interpolated frames/sec + ms/flow-calc, 2160p HDR, 1/2/4/8 MI355X",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed_max / a.steps, 4),
            "timed_region_s": round(elapsed_max, 3),
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue_s / a.steps, 4),
            "host_enqueue_note": f"host time of the calls of one step, measured on a burst of {BURST} periods issued into empty queues behind the "
                                 "timed region; host_issue_wall_ms_per_step is the wall time the issuing loop took INSIDE the timed region, "
                                 "which includes blocking on full hardware queues (back-pressure of the GPU), not host work",
            "host_issue_wall_ms_per_step": round(1e3 * host_issue_wall_s / a.steps, 4),
            "higher_is_better": True, "scaling": "strong" if a.workload in TOTAL_PAIRS else "weak", "vs_baseline": None,
            "dtype": "u16" if hdr else "u8", "data": "synthetic",
            "config": {"workload": a.workload, "description": desc, "scene": a.scene, "pool_frames": a.pool, "pool_order": a.pool_order, "search_radius": a.radius,
                       "delta_scalar": 8, "neighbor_scalar": a.neighbor, "blur_radius": a.blur_radius,
                       "pair_streams_per_gpu": a.streams, "pair_streams_total": a.streams * n_gpus,
                       "host_calls_per_batch_and_period": 1 if one_call else 3, "flow_batch": a.batch, "batch_streams_per_gpu": a.streams // a.batch,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "rank_devices": rank_devices, "device_count": torch.cuda.device_count(),
                       "launches_per_batch_and_period": ("1 phase-plane build + 12-launch chain graph + 1 fused warp" if a.eager_planes or not deferred_planes else "1 grid-sample launch + 1 fused warp (also builds the phase planes of frame N-1) + 12-launch chain graph") if (batches or a.batch > 1) and not a.member_warps else "per member",
                       "source_frames": "copied into the ring" if a.copy_in else "referenced in place (zero-copy)",
                       "source_periods_per_step": a.streams * P, "source_periods_per_stream_and_step": P,
                       "output_frames_total": int(frames_total), "parallelism": f"pair-sharded x{n_gpus}, no collective",
                       "frame_bytes": F, "flow_grid": [st["low_width"], st["low_height"]], "res_scalar": st["res_scalar"]},
            "device": dict(device_block(dev_index, sysfs, dev_mid[0] if dev_mid else None, dev_mid[-1] if dev_mid else None),
                           shader_clock_mhz_under_load=clock_under_load, clock_probe_call_ms=clock_probe_call_ms, hbm_streams_idle_device=hbm_idle,
                           shader_clock_note="hf_clock_probe behind the last warm-up step (same load, outside the timed region): shader cycles per 100 MHz reference tick "
                                             "over 3 ms, one wave beside the running pipeline -- the clock the device actually sustains under this load (pp_dpm sclk is the level requested)"),
            "ms_per_flow_calc": round(prof["flow_ms"] / prof["flow_chains"], 4) if prof["flow_chains"] else None,
            "ms_per_flow_calc_note": "device time of one refinement chain + blur while the other batch streams keep the GPU busy"
                                     + (f"; chains run {a.batch} pairs per launch (hf_batch): this is the batch's time / {a.batch}" if a.batch > 1 else ""),
            "ms_per_flow_calc_isolated": round(isolated["flow_chain_us"] / 1e3, 4) if isolated else None,
            "ms_per_flow_calc_isolated_after_sync": round(isolated["flow_chain_after_sync_us"] / 1e3, 4) if isolated else None,
            "ms_per_flow_calc_isolated_cache_warm": round(isolated["flow_chain_warm_us"] / 1e3, 4) if isolated else None,
            "ms_per_flow_calc_isolated_note": "one pair's refinement chain + blur alone on the GPU, device time per chain, every chain on a NEW frame right behind that frame's plane "
                                              "build (its plane rows come from HBM): 48 calls back to back / 12 calls each behind a host synchronisation (what rounds 1-4 printed under "
                                              "the first name); _cache_warm: 48 calls on ONE frame pair, planes cache-warm -- the way reference_opencl.ms_per_flow_calc is taken "
                                              "(= tools/chain_time.py --batch 1)",
            "roofline": roof,
        }
        if host_io_ranks is not None:
            ok = [r["async_pinned_side_streams"] for r in host_io_ranks if r and "async_pinned_side_streams" in r]
            out["host_io"] = {
                "ranks_reporting": len(ok), "aggregate_frames_per_s": round(sum(r["frames_per_s"] for r in ok), 1),
                "d2h_GB_per_s_per_gpu": round(sum(r["d2h_GB_per_s"] for r in ok) / max(len(ok), 1), 2),
                "h2d_GB_per_s_per_gpu": round(sum(r["h2d_GB_per_s"] for r in ok) / max(len(ok), 1), 2),
                "per_rank_frames_per_s": [r["frames_per_s"] for r in ok],
                "errors": [r.get("error") for r in host_io_ranks if r and "error" in r],
                "note": "every rank at the same time: one asynchronous context per GPU (child process), pinned host buffers, H2D / D2H on side "
                        "streams, every output frame returned to the host (PCIe-inclusive; never the bench `value`)"}
        if content_counters:
            out["content_counters"] = content_counters
        if world == 1 and a.workload == "hdr2160_24to120" and not a.no_other_workloads:
            out["other_workloads"] = other_workloads(a)
        if world == 1 and a.workload == "hdr2160_24to120" and a.scene == "bench" and not a.no_content_legs:
            out["content"] = content_legs(a)
        if not a.no_host_io and world == 1:
            try:
                out["host_io"] = host_io_block(hdr, H, W, target)
            except Exception as e:
                out["host_io"] = {"error": repr(e)}
        if not a.no_cpu_baseline and world == 1:   # reported baselines: rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(hdr, H, W, target, a.radius, a.neighbor, host_frames, a.cpu_sample_pairs)
                out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(hdr, H, W, target, a.radius, a.neighbor, host_frames)
            except Exception as e:  # the checker must never take the measurement down
                out["cpu_baseline"] = {"error": repr(e)}
        if not a.no_reference and world == 1:
            try:
                r = reference_opencl(hdr, H, W, target, a.radius, a.neighbor, host_frames)
                if r:
                    out["reference_opencl"] = r
                    out["speedup_vs_reference_opencl"] = round(out["value"] / n_gpus / r["frames_per_s"], 2)
            except Exception as e:
                out["reference_opencl"] = {"error": repr(e)}
        if timeline:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from timeline_report import analyze
            out["DIAGNOSTIC_NOT_A_BENCHMARK"] = "timeline: the chains of %.2g steps were issued launch by launch with events on every dispatch" % a.timeline_steps
            run = {"frames_per_s": out["value"], "ms_per_step": out["ms_per_step"], "workload": a.workload, "batch_streams": a.streams // a.batch,
                   "flow_batch": a.batch, "steps": a.steps, "timeline_steps": a.timeline_steps, "device": out.get("device")}
            rep = analyze(timeline["streams"], timeline["alone"], P)
            rep["run"] = run
            os.makedirs(os.path.dirname(os.path.abspath(a.timeline_out)), exist_ok=True)
            json.dump(rep, open(a.timeline_out, "w"), indent=1)
            json.dump(dict(timeline, run=run), open(a.timeline_out.replace(".json", "") + "_raw.json", "w"))
            out["timeline"] = {"file": a.timeline_out, "implied_ms_per_step": rep.get("implied_ms_per_step"), "mean_queues_busy": (rep.get("concurrency") or {}).get("mean_queues_busy"),
                               "serial_over_pipelined": rep.get("serial_over_pipelined")}
        result_line = json.dumps(out)

    for b in batches:
        b.close()
    for c in calcs:
        c.close()
    if world > 1 or a.pg_first:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL prints a version banner through C stdio, which would otherwise be flushed
        # behind it at exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
