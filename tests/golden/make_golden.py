"""tests/golden/make_golden.py -- generates the committed golden vectors.

Runs THE REFERENCE ITSELF (oracle/_ref/ref_runner = the unmodified reference host code and
OpenCL kernels, built by oracle/Makefile from /root/reference in the build container) on an
MI355X through the AMD OpenCL runtime, on seeded synthetic inputs (hopperrender_amd/synth.py),
replaying the filter's call protocol (reference HopperRender.cpp:953-957,1179-1186):

    update f0, f1, f2 ; calc            -> offsets, blurred flow (f1->f2), totalFrameDelta
    update f3 ; calc                    -> offsets, blurred flow (f2->f3)
    warp(t, mode) ; download            -> output frames from f1, f2 and flow (f1->f2)
    copy ; download

Usage (on the GPU box, from the repo root):
    python tests/golden/make_golden.py [--out gpurun_out/golden] [--big]
The results are written as .npz (small cases: raw arrays; big cases: SHA-256 + strided probes)
and then copied by hand into tests/golden/.  Inputs are NOT stored: they are regenerated from
the seed; a SHA-256 of every input frame is stored to detect generator drift.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hopperrender_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

T_VALUES = [0.0, 0.1988, 0.3996, 0.5, 0.7992, 0.998, 1.0]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ramp_frame(H, W, hdr, stride, phase=0):
    """Code ramp: Y plane holds every code value once H*W >= 65536 (HDR) / 256 (SDR); the UV plane
    holds every code over phases 0 and 1 (levels tests)."""
    n = (H + H // 2) * stride
    a = np.arange(n, dtype=np.uint32)
    a[H * stride:] += phase * (n - H * stride)
    if hdr:
        return (a % 65536).astype(np.uint16)
    return (a % 256).astype(np.uint8)


CASES = [
    # name, hdr, H, W, in_stride, out_stride, seed, list of (R, delta, neighbor), content
    dict(name="sdr_180p", hdr=0, H=180, W=320, si=0, so=0, seed=11, params=[(5, 8, 6), (8, 8, 6), (16, 8, 6), (16, 4, 10)], raw=True),
    dict(name="hdr_180p", hdr=1, H=180, W=320, si=0, so=0, seed=12, params=[(16, 8, 6)], raw=True),
    dict(name="sdr_360p", hdr=0, H=360, W=640, si=0, so=0, seed=1234, params=[(5, 8, 6), (16, 8, 6)]),
    dict(name="hdr_360p", hdr=1, H=360, W=640, si=0, so=0, seed=1235, params=[(5, 8, 6), (16, 8, 6), (9, 10, 0)]),
    dict(name="sdr_ragged_strided", hdr=0, H=338, W=600, si=640, so=608, seed=21, params=[(7, 8, 6), (16, 8, 6)]),
    dict(name="hdr_ragged_strided", hdr=1, H=338, W=600, si=608, so=640, seed=22, params=[(6, 8, 6), (16, 6, 3)]),
    dict(name="sdr_722p_rs2", hdr=0, H=722, W=1282, si=0, so=0, seed=31, params=[(16, 8, 6)], big=True),
    dict(name="sdr_identical", hdr=0, H=180, W=320, si=0, so=0, seed=41, params=[(16, 8, 6)], content="identical"),
    dict(name="sdr_scenecut", hdr=0, H=180, W=320, si=0, so=0, seed=42, params=[(8, 8, 6)], content="cut"),
    dict(name="sdr_1080p", hdr=0, H=1080, W=1920, si=0, so=0, seed=1234, params=[(5, 8, 6), (16, 8, 6)], big=True),
    dict(name="hdr_2160p", hdr=1, H=2160, W=3840, si=0, so=0, seed=1234, params=[(16, 8, 6), (16, 8, 10)], big=True),
    # maxCalcRes above the default: no down-scaling, a 1040-wide grid whose levels with the neighbour term still have windows > 32
    dict(name="sdr_widegrid_rs0", hdr=0, H=200, W=1040, si=0, so=0, seed=51, params=[(12, 3, 4), (16, 8, 6)], big=True, max_res=1000),
]


def frames_for(case):
    sc = synth.Scene(case["H"], case["W"], bool(case["hdr"]), case["seed"], in_stride=case["si"])
    content = case.get("content", "motion")
    if content == "identical":
        f = sc.frame(0)
        return [f, f.copy(), f.copy(), f.copy()]
    if content == "cut":
        other = synth.Scene(case["H"], case["W"], bool(case["hdr"]), case["seed"] + 999, in_stride=case["si"])
        return [sc.frame(0), sc.frame(1), other.frame(2), other.frame(3)]
    return [sc.frame(k) for k in range(4)]


def run_case(case, outdir, modes_small=(0, 1, 2, 3, 4, 5, 6)):
    big = case.get("big", False)
    raw = case.get("raw", False)
    frames = frames_for(case)
    res = {"meta": dict(case), "inputs_sha": [sha(f) for f in frames]}
    arrays = {}
    for (R, delta, nb) in case["params"]:
        key = f"R{R}_d{delta}_n{nb}"
        s = oracle.RefSession(case["hdr"], case["H"], case["W"], case["si"], case["so"], delta, nb, 0.0, 255.0, case.get("max_res", 270))
        g = s.g
        s.radius(R)
        for f in frames[:3]:
            s.update(f)
        s.calc(); s.stats()
        s.dump_offsets("off_a"); s.dump_blurred(1, "blur_a")
        s.update(frames[3]); s.radius(R)
        s.calc(); s.stats()
        s.dump_offsets("off_b"); s.dump_blurred(1, "blur_b"); s.dump_blurred(0, "blur_prev")
        outs = []
        tvals = T_VALUES if not big else [0.3996, 0.7992]
        modes = modes_small if not big else (0, 1, 2)
        for m in modes:
            for t in (tvals if m in (0, 1, 2) else [0.3996]):
                s.warp(t, m); s.download(f"warp_m{m}_t{t}")
                outs.append((m, t))
        s.copy(); s.download("copy_default")
        s.params(delta, nb, 16.0, 235.0)
        s.warp(0.5, 2); s.download("warp_m2_t0.5_lv16_235")
        s.copy(); s.download("copy_lv16_235")
        s.params(delta, nb, 0.0, 200.0)
        s.copy(); s.download("copy_lv0_200")
        t0 = time.time()
        js, arrs = s.run()
        dt = time.time() - t0
        assert (arrs["blur_prev"] == arrs["blur_a"]).all(), "ping-pong protocol broken"
        entry = {"stats_a": js[0], "stats_b": js[1], "geom": dict(rs=g.rs, lw=g.lw, lh=g.lh), "warps": outs,
                 "ref_wall_s": dt}
        # quick oracle comparison for immediate feedback
        off_o, blur_o, tot_o, oob = oracle.calculate_optical_flow(frames[1], frames[2], g, R, 0, delta, nb, 4)
        entry["oracle_check"] = dict(off_mismatch=int((off_o != arrs["off_a"]).sum()),
                                     blur_mismatch=int((blur_o != arrs["blur_a"]).sum()),
                                     total_delta_oracle=int(tot_o), oob=int(oob))
        wchk = {}
        for (m, t) in outs:
            o = oracle.warp_frames(frames[1], frames[2], arrs["blur_a"], g, t, m, 0.0, 255.0)
            r = arrs[f"warp_m{m}_t{t}"]
            d = np.abs(o.astype(np.int32) - r.astype(np.int32))
            wchk[f"m{m}_t{t}"] = [int((d != 0).sum()), int(d.max())]
        o = oracle.copy_frame(frames[1], g, 0.0, 255.0)
        d = np.abs(o.astype(np.int32) - arrs["copy_default"].astype(np.int32)); wchk["copy_default"] = [int((d != 0).sum()), int(d.max())]
        o = oracle.copy_frame(frames[1], g, 16.0, 235.0)
        d = np.abs(o.astype(np.int32) - arrs["copy_lv16_235"].astype(np.int32)); wchk["copy_lv16_235"] = [int((d != 0).sum()), int(d.max())]
        o = oracle.warp_frames(frames[1], frames[2], arrs["blur_a"], g, 0.5, 2, 16.0, 235.0)
        d = np.abs(o.astype(np.int32) - arrs["warp_m2_t0.5_lv16_235"].astype(np.int32)); wchk["warp_lv16_235"] = [int((d != 0).sum()), int(d.max())]
        entry["oracle_warp_check"] = wchk
        worst = max(v[1] for v in wchk.values())
        exact = sorted(k for k, v in wchk.items() if v[0] == 0)
        print(case["name"], key, json.dumps(entry["oracle_check"]), "warp/copy max|d|", worst,
              "exact:", " ".join(exact), flush=True)
        res[key] = entry
        for name, a in arrs.items():
            if name == "blur_prev":
                continue
            is_frame = name.startswith("warp") or name.startswith("copy")
            if is_frame:
                res[key].setdefault("sha", {})[name] = sha(a)
                if raw and (R, delta, nb) == case["params"][-1 if case["name"] == "hdr_180p" else 2]:
                    arrays[f"{key}/{name}"] = a
                else:
                    arrays[f"{key}/{name}/probe"] = a[::997].copy()
            else:
                arrays[f"{key}/{name}"] = a
    np.savez_compressed(os.path.join(outdir, case["name"] + ".npz"), meta=json.dumps(res), **arrays)
    return res


def levels_case(outdir):
    """Every code value through copyFrame at several level settings (copyFrameKernel{SDR,HDR}.h)."""
    arrays, meta = {}, {}
    for hdr in (0, 1):
        H, W = 256, 256
        for phase in (0, 1):
            f = ramp_frame(H, W, hdr, W, phase)
            for (bk, wh) in [(0.0, 255.0), (16.0, 235.0), (0.0, 200.0), (30.0, 180.0), (0.0, 128.0), (7.0, 251.0)]:
                s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, bk, wh, 270)
                s.update(f); s.copy(); s.download("c")
                _, arrs = s.run()
                name = f"{'hdr' if hdr else 'sdr'}_p{phase}_b{int(bk)}_w{int(wh)}"
                arrays[name] = arrs["c"]
                o = oracle.copy_frame(f, s.g, bk, wh)
                d = np.abs(o.astype(np.int32) - arrs["c"].astype(np.int32))
                meta[name] = [int((d != 0).sum()), int(d.max())]
                print("levels", name, meta[name], flush=True)
    np.savez_compressed(os.path.join(outdir, "levels_ramp.npz"), meta=json.dumps(meta), **arrays)


def timing(outdir):
    """Wall-clock of the reference's own OpenCL path on this GPU (baseline record, not a fixture)."""
    rows = []
    for (name, hdr, H, W) in [("sdr_1080p", 0, 1080, 1920), ("hdr_2160p", 1, 2160, 3840)]:
        sc = synth.Scene(H, W, bool(hdr), 1234)
        fr = [sc.frame(k) for k in range(4)]
        for R in (5, 16):
            s = oracle.RefSession(hdr, H, W, 0, 0, 8, 6, 0.0, 255.0, 270)
            s.radius(R)
            for f in fr:
                s.update(f)
            s.calc(); s.calc()
            s.time_calc(200); s.time_warp(200, 0.3996, 2); s.stats()
            js, _ = s.run()
            row = dict(case=name, R=R, time_calc_ms=js[0]["time_calc_ms"], time_warp_ms=js[1]["time_warp_ms"],
                       ofc_calc_time_s=js[2]["ofc_calc_time"])
            rows.append(row); print("timing", json.dumps(row), flush=True)
    json.dump(rows, open(os.path.join(outdir, "reference_opencl_timing.json"), "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "golden"))
    ap.add_argument("--only", default="")
    ap.add_argument("--no-timing", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if not oracle.ref_available():
        sys.exit("needs oracle/_ref/ref_runner and an OpenCL GPU")
    for c in CASES:
        if a.only and a.only not in c["name"]:
            continue
        run_case(c, a.out)
    if not a.only:
        levels_case(a.out)
        if not a.no_timing:
            timing(a.out)


if __name__ == "__main__":
    main()
