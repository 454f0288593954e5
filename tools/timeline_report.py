"""tools/timeline_report.py -- what the batch streams of the pipeline do at the same time, from the per-dispatch device timestamps of
hf_batch_timeline_* (no profiler attached; include/hopperflow.h).

    analyze(streams, alone=None, periods_per_step=64) -> dict            (pure; tests/test_timeline_report.py)
    python tools/timeline_report.py RAW.json [--out profiles/rNN_pipeline_timeline.json]

`streams`: one list per batch stream of (kernel, period, start_ms, end_ms), all on one clock (one reference event per process and device).
`alone`: the same records of ONE batch stream running with the others idle (stand-alone durations of the same launches).

Only the window in which EVERY stream is being recorded is analysed (from the latest first start to the earliest last end).  Per queue:
busy time (sum of its dispatches), idle gaps between consecutive dispatches, periods finished; per kernel: duration inside the pipeline,
stand-alone duration and the stretch factor between them, share of the queues' busy time, and the concurrency histogram -- for how much
of its run time 0, 1, 2 ... OTHER queues had a dispatch running, and how many of those were period warps."""
import argparse
import json
import os
import statistics


def _clip(recs, t0, t1):
    out = []
    for k, p, s, e in recs:
        s2, e2 = max(s, t0), min(e, t1)
        if e2 > s2:
            out.append((k, p, s2, e2, s, e))
    return out


def _kernel_stats(recs):
    by = {}
    for k, _, s, e in recs:
        by.setdefault(k, []).append((e - s) * 1e3)
    return {k: {"n": len(v), "mean_us": round(statistics.fmean(v), 2), "median_us": round(statistics.median(v), 2),
                "min_us": round(min(v), 2), "max_us": round(max(v), 2)} for k, v in by.items()}


def analyze(streams, alone=None, periods_per_step=64, warp_name="warp_period"):
    streams = [sorted(s, key=lambda r: r[2]) for s in streams if s]
    if not streams:
        return {"error": "no records"}
    t0 = max(s[0][2] for s in streams)
    t1 = min(s[-1][3] for s in streams)
    if t1 <= t0:
        return {"error": "the streams' recorded windows do not overlap", "windows_ms": [[s[0][2], s[-1][3]] for s in streams]}
    span = t1 - t0
    clipped = [_clip(s, t0, t1) for s in streams]

    # ---- per queue ----
    queues = []
    for qi, recs in enumerate(clipped):
        busy = sum(e - s for _, _, s, e, _, _ in recs)
        gaps = [(recs[i + 1][2] - recs[i][3]) * 1e3 for i in range(len(recs) - 1)]
        gaps = [max(g, 0.0) for g in gaps]
        periods = sorted({p for k, p, s, e, s0, e0 in recs if s0 >= t0 and e0 <= t1})
        whole = [p for p in periods if all(s0 >= t0 and e0 <= t1 for k, pp, s, e, s0, e0 in recs if pp == p)]
        # period time of this queue: from the first dispatch of one period to the first dispatch of the next
        firsts = {}
        for k, p, s, e, s0, e0 in recs:
            firsts[p] = min(firsts.get(p, s0), s0)
        ps = sorted(firsts)
        per = [(firsts[ps[i + 1]] - firsts[ps[i]]) for i in range(len(ps) - 1) if ps[i + 1] == ps[i] + 1]
        queues.append({
            "queue": qi, "dispatches": len(recs), "busy_ms": round(busy, 4), "busy_frac": round(busy / span, 4),
            "idle_ms": round(span - busy, 4), "periods_in_window": len(whole),
            "mean_period_ms": round(statistics.fmean(per), 5) if per else None,
            "implied_ms_per_step": round(statistics.fmean(per) * periods_per_step, 3) if per else None,
            "gap_us": {"mean": round(statistics.fmean(gaps), 2) if gaps else None, "median": round(statistics.median(gaps), 2) if gaps else None,
                       "max": round(max(gaps), 2) if gaps else None, "sum_ms": round(sum(gaps) / 1e3, 4)},
        })

    # ---- sweep: who runs beside whom ----
    events = []
    for qi, recs in enumerate(clipped):
        for k, p, s, e, _, _ in recs:
            events.append((s, 1, qi, k))
            events.append((e, 0, qi, k))
    events.sort(key=lambda x: (x[0], x[1]))     # ends before starts at equal times
    running = {}                                  # queue -> kernel
    nq = len(clipped)
    conc_time = [0.0] * (nq + 1)                  # time with n queues busy
    beside = {}                                   # kernel -> {"others": [time with n other queues busy], "other_warps": [...], "total": t}
    last = t0
    for t, is_start, qi, k in events:
        dt = t - last
        if dt > 0:
            n = len(running)
            conc_time[n] += dt
            n_warp = sum(1 for v in running.values() if v == warp_name)
            for q, kk in running.items():
                b = beside.setdefault(kk, {"others": [0.0] * nq, "other_warps": [0.0] * nq, "total": 0.0})
                b["others"][n - 1] += dt
                b["other_warps"][n_warp - (1 if kk == warp_name else 0)] += dt
                b["total"] += dt
        last = t
        if is_start:
            running[qi] = k
        else:
            running.pop(qi, None)
    conc = {"time_frac_with_n_queues_busy": [round(x / span, 4) for x in conc_time],
            "mean_queues_busy": round(sum(i * x for i, x in enumerate(conc_time)) / span, 3)}

    # ---- per kernel ----
    inside = _kernel_stats([(k, p, s0, e0) for recs in clipped for k, p, s, e, s0, e0 in recs if s0 >= t0 and e0 <= t1])
    alone_stats = _kernel_stats(alone) if alone else {}
    total_busy = sum(q["busy_ms"] for q in queues)
    kernels = {}
    for k, st in sorted(inside.items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["n"]):
        b = beside.get(k, {"others": [0.0] * nq, "other_warps": [0.0] * nq, "total": 0.0})
        tot = b["total"] or 1.0
        ent = dict(st)
        ent["share_of_busy_time"] = round(st["mean_us"] * st["n"] / 1e3 / total_busy, 4) if total_busy else None
        ent["time_frac_with_n_other_queues_busy"] = [round(x / tot, 4) for x in b["others"]]
        ent["time_frac_with_n_other_period_warps_running"] = [round(x / tot, 4) for x in b["other_warps"]]
        ent["mean_other_queues_busy"] = round(sum(i * x for i, x in enumerate(b["others"])) / tot, 3)
        if k in alone_stats:
            ent["alone_mean_us"] = alone_stats[k]["mean_us"]
            ent["alone_n"] = alone_stats[k]["n"]
            ent["stretch_vs_alone"] = round(st["mean_us"] / alone_stats[k]["mean_us"], 3) if alone_stats[k]["mean_us"] else None
        kernels[k] = ent
    # the pipeline's kernel time if nothing overlapped, and what the overlap gives
    sum_alone_per_period = None
    if alone_stats and queues and queues[0]["periods_in_window"]:
        per_period = {}
        for k, st in inside.items():
            per_period[k] = st["n"] / max(1, sum(q["periods_in_window"] for q in queues))
        sum_alone_per_period = sum(alone_stats[k]["mean_us"] * per_period[k] for k in inside if k in alone_stats)
    mean_period = statistics.fmean([q["mean_period_ms"] for q in queues if q["mean_period_ms"]]) if any(q["mean_period_ms"] for q in queues) else None
    out = {
        "window_ms": round(span, 4), "queues": queues, "concurrency": conc, "kernels": kernels,
        "sum_of_queue_busy_ms": round(total_busy, 4),
        "mean_period_ms_per_queue": round(mean_period, 5) if mean_period else None,
        "implied_ms_per_step": round(mean_period * periods_per_step, 3) if mean_period else None,
    }
    if sum_alone_per_period and mean_period:
        out["alone_kernel_time_per_batch_period_us"] = round(sum_alone_per_period, 1)
        out["serial_over_pipelined"] = round(sum_alone_per_period * nq / 1e3 / (mean_period * 1.0) / 1.0, 3)   # nq queues' stand-alone time per period of wall time
        out["serial_over_pipelined_note"] = ("(stand-alone kernel time of one batch period x queues) / (wall time in which every queue finishes one period): "
                                             "1.0 = the queues gain nothing from running side by side")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("raw")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    raw = json.load(open(a.raw))
    rep = analyze(raw["streams"], raw.get("alone"), raw.get("periods_per_step", 64))
    rep["run"] = raw.get("run")
    txt = json.dumps(rep, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
