"""tools/pmc_traffic.py -- profiles/roofline_traffic.json from rocprofv3 PMC passes (generate it, do not type it).

    python tools/pmc_traffic.py <workload> <fetch_csv> <write_csv> [--out profiles/roofline_traffic.json]

<fetch_csv> / <write_csv>: the *_counter_collection.csv of two separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
passes over the same command (bench.py --streams 1 --batch 1 --no-profile: one fused period launch at a time).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": the counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads (16 B per lane) -> doubled; WRITE_SIZE is exact for 16-byte streaming stores.
The entry describes the dominant kernel: the fused period launch with the most output frames."""
import argparse
import collections
import csv
import json
import os


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def short(name):
    """'void hf::(anonymous namespace)::flow_level_small_kernel<16, true>(hf::Geom, hf::(anonymous namespace)::FlowBatchArgs)'
    -> 'flow_level_small_kernel<16, true>'"""
    n = name[5:] if name.startswith("void ") else name
    depth = 0
    for i, ch in enumerate(n):               # cut the argument list: the first '(' outside template brackets that opens it
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0 and not n[i:].startswith("(anonymous namespace)"):
            n = n[:i]
            break
    return n.replace("hf::(anonymous namespace)::", "").replace("hf::", "")


def pipeline_entry(a):
    """HBM traffic of EVERY kernel inside the batched pipeline, per pair and source period and per output frame.
    Inputs: PMC passes (FETCH_SIZE, WRITE_SIZE) over the default `bench.py` command (its operating point: batches of
    `--batch` members).  Only the batched dispatches count (per kernel: the dispatches with the largest grid; priming calls
    and the stand-alone leg behind the timed region launch one member at a time).  A batched warp dispatch = one source
    period of `--batch` pairs; output frames = bytes it wrote / frame bytes."""
    f_all, w_all = per_kernel(a.fetch_csv, "FETCH_SIZE"), per_kernel(a.write_csv, "WRITE_SIZE")

    def batched(acc):
        by_name = collections.defaultdict(dict)
        for (name, grid), v in acc.items():
            by_name[name][grid] = v
        return {name: grids[max(grids)] for name, grids in by_name.items() if "hf::" in name}

    fb, wb = batched(f_all), batched(w_all)
    warp = [n for n in wb if a.kernel in n]
    if not warp:
        raise SystemExit("no %s dispatches" % a.kernel)
    warp = max(warp, key=lambda n: sum(wb[n]))
    periods_w, periods_f = len(wb[warp]) * a.batch, len(fb[warp]) * a.batch      # pair-periods seen by each pass
    # output frames per pair and period: the workload's schedule (source / target frame time) when given -- the warp launch may also
    # write the phase planes of a frame (deferred build), so its WRITE_SIZE is not only output frames
    frames_per_pair_period = a.outputs_per_period if a.outputs_per_period > 0 else sum(wb[warp]) * 1024 / a.frame_bytes / periods_w
    per_kernel_bytes, total = {}, 0.0
    for name in sorted(set(fb) | set(wb)):
        rd = 2 * sum(fb.get(name, [])) * 1024 / periods_f                          # FETCH_SIZE x 2 (gfx950 note), KiB -> bytes
        wr = sum(wb.get(name, [])) * 1024 / periods_w
        if rd + wr < 1:
            continue
        # kernels that only the single-member priming calls launch (e.g. the row-per-lane level kernels of batches <= 4) are not the pipeline
        if len(wb.get(name, [])) < 0.5 * len(wb[warp]) and len(fb.get(name, [])) < 0.5 * len(fb[warp]):
            continue
        k = short(name)
        assert k not in per_kernel_bytes, k
        per_kernel_bytes[k] = {"read": int(rd), "write": int(wr), "dispatches_per_pair_period": round(len(wb.get(name, [])) / periods_w * a.batch, 2)}
        total += rd + wr
    return {
        "hbm_bytes_per_output_frame": int(total / frames_per_pair_period),
        "hbm_bytes_per_pair_and_period": int(total),
        "output_frames_per_pair_and_period": round(frames_per_pair_period, 3),
        "per_kernel_bytes_per_pair_and_period": per_kernel_bytes,
        "operating_point": "%d pairs per batch, batched dispatches only" % a.batch,
        "pair_periods": {"fetch_pass": periods_f, "write_pass": periods_w},
        "corrections": "FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests of wide coalesced loads at 64 B; an upper bound for the narrow "
                       "gathers of the chain kernels), KiB -> bytes; WRITE_SIZE exact; counter collection serialises the kernels, so the "
                       "cross-stream cache effects of the free-running pipeline are not in these numbers",
        "files": [os.path.relpath(a.fetch_csv), os.path.relpath(a.write_csv)],
    }


def git_commit():
    """Short hash of HEAD when the record is generated (the kernels whose bytes it holds); bench.py prints it beside roofline.frac."""
    import subprocess
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        h = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "hopperrender_amd/csrc"], capture_output=True, text=True).stdout.strip()
        return h + ("+uncommitted-kernel-changes" if dirty else "")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload"); ap.add_argument("fetch_csv"); ap.add_argument("write_csv")
    ap.add_argument("--pipeline", action="store_true", help="the CSVs are passes over the batched pipeline: write the workload's `pipeline` entry")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "roofline_traffic.json"))
    ap.add_argument("--kernel", default="::warp_", help="substring of the dominant kernel's name (warp_fast_kernel / warp_wg_kernel)")
    ap.add_argument("--units-per-launch", type=float, default=0.0, help="output frames of the launches selected (0: take it from --frames-by-write)")
    ap.add_argument("--outputs-per-period", type=float, default=0.0, help="--pipeline: output frames per pair and source period of the workload's schedule")
    ap.add_argument("--frame-bytes", type=int, default=0, help="bytes of one output frame: units per launch = WRITE_SIZE / frame bytes")
    a = ap.parse_args()
    if a.pipeline:
        if not a.frame_bytes:
            raise SystemExit("--pipeline needs --frame-bytes")
        data = json.load(open(a.out)) if os.path.exists(a.out) else {}
        data.setdefault(a.workload, {})["pipeline"] = pipeline_entry(a)
        data[a.workload]["git_commit"] = git_commit()
        json.dump(data, open(a.out, "w"), indent=1)
        print(json.dumps({a.workload: {"pipeline": data[a.workload]["pipeline"]}}, indent=1))
        return
    f = {k: v for k, v in per_kernel(a.fetch_csv, "FETCH_SIZE").items() if a.kernel in k[0]}
    w = {k: v for k, v in per_kernel(a.write_csv, "WRITE_SIZE").items() if a.kernel in k[0]}
    if not f or not w:
        raise SystemExit("no %s dispatches in the CSVs" % a.kernel)
    # several launch shapes of the same kernel may appear (5- and 6-output periods share a grid): bucket by written bytes
    name, grid = max(w, key=lambda k: sum(w[k]) / len(w[k]))
    wv, fv = w[(name, grid)], f[(name, grid)]
    units = None
    if a.frame_bytes:
        by_units = collections.defaultdict(list)
        for x in wv:
            by_units[round(x * 1024 / a.frame_bytes)].append(x)
        units = max(by_units)                      # the launches with the most output frames
        wv = by_units[units]
    write_kib = sum(wv) / len(wv)
    fetch_kib = sum(fv) / len(fv)
    entry = {
        "warp_kernel_hbm_bytes_per_launch": int(round((2 * fetch_kib + write_kib) * 1024)),
        "units_per_launch": units if units is not None else a.units_per_launch,
        "kernel": name, "grid_size": grid,
        "FETCH_SIZE_KiB_mean": round(fetch_kib, 1), "WRITE_SIZE_KiB_mean": round(write_kib, 1),
        "dispatches": {"fetch_pass": len(fv), "write_pass": len(wv)},
        "corrections": "FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests of wide coalesced loads at 64 B), KiB -> bytes; WRITE_SIZE exact",
        "files": [os.path.relpath(a.fetch_csv), os.path.relpath(a.write_csv)],
        "note": "FETCH_SIZE is averaged over all launches of the shape (5- and 6-output periods read the same two source frames); "
                "WRITE_SIZE over the launches with the most output frames",
    }
    data = {}
    if os.path.exists(a.out):
        try:
            data = json.load(open(a.out))
        except Exception:
            data = {}
    if "pipeline" in data.get(a.workload, {}):
        entry["pipeline"] = data[a.workload]["pipeline"]
    entry["git_commit"] = git_commit()
    data[a.workload] = entry
    json.dump(data, open(a.out, "w"), indent=1)
    print(json.dumps({a.workload: entry}, indent=1))


if __name__ == "__main__":
    main()
