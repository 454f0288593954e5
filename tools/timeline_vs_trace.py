"""tools/timeline_vs_trace.py RAW_TIMELINE.json KERNEL_TRACE.csv -- the same dispatches seen twice, in ONE process: by the library's own
events (hf_batch_timeline_*: hipExtLaunchKernelGGL start / stop events, include/hopperflow_diag.h) and by rocprofv3 --kernel-trace
(start / end of the dispatch's completion signal).  Run as
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --timeline-out RAW.json ...
The timeline's records are matched to the trace's rows per queue by their END times (both are completion timestamps; the clocks differ by
one constant, found from the first matched kernel), and the report says how the START times differ per kernel: what a start event measures."""
import collections
import csv
import json
import statistics
import sys

LABEL = (("warp_wg_kernel", "warp_period"), ("warp_fast_kernel", "warp_period"), ("prep_grid_kernel", "grid_samples"), ("prep_phase_fast_kernel", "plane"),
         ("flow_big_partial_kernel", "large_windows"), ("flow_level_small_kernel<32", "level_32"), ("flow_level32_wave_kernel", "level_32"), ("flow_level_small_kernel<16", "level_16"),
         ("flow_level_small_kernel<8", "level_8"), ("flow_level_small_kernel<4", "level_4"), ("flow_level_small_kernel<2", "level_2"), ("blur_flow_kernel", "blur"))


def label(name):
    for pat, lab in LABEL:
        if pat in name:
            return lab
    return None


def main(raw_path, trace_path):
    raw = json.load(open(raw_path))
    streams = [sorted(s, key=lambda r: r[3]) for s in raw["streams"] if s]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), label(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(trace_path))]
    rows = [r for r in rows if r[2]]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    for q in byq:
        byq[q].sort(key=lambda r: r[1])
    print("# event view (library) vs kernel trace (rocprofv3), same process, dispatches matched by completion time")
    for si, recs in enumerate(streams):
        labs = [r[0].replace("_x", "").replace("_y", "") if r[0].startswith("large_windows") else r[0] for r in recs]
        ends = [r[3] * 1e6 for r in recs]          # ns on the event clock
        best = None
        for q, tr in byq.items():
            tl = [r[2] for r in tr]
            # the trace holds the whole run: slide the timeline's label sequence over it, keep offsets where labels AND end-to-end spacings agree
            n = len(labs)
            first = labs[0]
            for off in range(0, len(tl) - n + 1):
                if tl[off] != first or tl[off:off + n] != labs:
                    continue
                d0 = tr[off][1] - ends[0]
                err = max(abs((tr[off + i][1] - ends[i]) - d0) for i in range(0, n, max(1, n // 50)))
                if best is None or err < best[0]:
                    best = (err, q, off, d0)
        if best is None:
            print("stream", si, ": no queue of the trace carries this label sequence"); continue
        err, q, off, d0 = best
        tr = byq[q][off:off + len(recs)]
        print("stream %d = trace queue %s, %d dispatches, completion times agree within %.1f us (clock offset removed)" % (si, q, len(recs), err / 1e3))
        per = collections.defaultdict(lambda: [[], [], []])
        for (name, _, s_ms, e_ms), (ts, te, lab, _) in zip(recs, tr):
            ev_d = (e_ms - s_ms) * 1e3; tr_d = (te - ts) / 1e3
            start_lag = ((ts - d0) - s_ms * 1e6) / 1e3          # trace start minus event start, us
            per[lab][0].append(ev_d); per[lab][1].append(tr_d); per[lab][2].append(start_lag)
        for lab, (a, b, c) in per.items():
            print("   %-14s n=%4d  event duration %8.1f us   trace duration %8.1f us   trace start - event start %8.1f us (median %8.1f)" %
                  (lab, len(a), statistics.fmean(a), statistics.fmean(b), statistics.fmean(c), statistics.median(c)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
