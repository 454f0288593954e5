"""tools/chain_beside_warp.py KERNEL_TRACE.csv -- from a rocprofv3 --kernel-trace of the default bench (tools/final_profiles.sh:
gpurun_out/<tag>/stats_default/p_kernel_trace.csv): how long the chain's launches take while a period warp of ANOTHER batch stream
runs beside them (>= 90 % of the launch overlapped) and while none does (< 10 %), and how many warp launches run at once."""
import collections, csv, re, statistics, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(warp_wg_kernel|flow_big_partial_kernel|flow_level32_wave_kernel|flow_level_small_kernel<\d+|blur_flow_kernel)", r["Kernel_Name"])
    if m:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), int(r["Queue_Id"]), int(r["Grid_Size_X"])))
rows.sort()
t0, t1 = rows[len(rows) // 5][0], rows[4 * len(rows) // 5][0]          # the middle of the run
sel = [r for r in rows if t0 <= r[0] <= t1]
warps = sorted((s, e, q) for s, e, n, q, _ in sel if n == "warp_wg_kernel")
ev = sorted([(s, 1) for s, _, _ in warps] + [(e, -1) for _, e, _ in warps])
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("warp launches running at once:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})


def beside(s, e, q):
    return sum(min(e, we) - max(s, ws) for ws, we, wq in warps if we > s and ws < e and wq != q) / (e - s)


for name in ("flow_big_partial_kernel", "flow_level32_wave_kernel", "flow_level_small_kernel<32", "flow_level_small_kernel<16", "flow_level_small_kernel<8",
             "flow_level_small_kernel<4", "flow_level_small_kernel<2"):
    ks = [r for r in sel if r[2] == name and r[4] > 20000][:3000]     # batched launches only
    f = [(beside(s, e, q), (e - s) / 1e3) for s, e, _, q, _ in ks]
    lo, hi = [d for b, d in f if b < 0.1], [d for b, d in f if b > 0.9]
    print("%-28s no warp beside it: n=%4d mean %6.1f us    a warp beside it: n=%4d mean %6.1f us" %
          (name, len(lo), statistics.mean(lo) if lo else 0, len(hi), statistics.mean(hi) if hi else 0))
