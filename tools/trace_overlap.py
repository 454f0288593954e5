"""tools/trace_overlap.py DIR -- concurrency / per-stream gap statistics from a rocprofv3 kernel trace (csv)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    r["n"] = re.sub(r"void |hf::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:26]
rows = [r for r in rows if "rocclr" not in r["n"]]
rows.sort(key=lambda r: r["s"])
n = len(rows)
win = rows[n // 3: 2 * n // 3]                       # steady-state third
T0, T1 = win[0]["s"], max(r["e"] for r in win)
print(f"window {(T1 - T0) / 1e3:.0f} us, {len(win)} kernels, queues {sorted(set(r['Queue_Id'] for r in win))}")
ev = []
for r in win:
    ev += [(r["s"], 1), (r["e"], -1)]
ev.sort()
c = 0; last = T0; hist = collections.Counter()
for t, d in ev:
    hist[c] += t - last; last = t; c += d
tot = sum(hist.values())
print("kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per-stream: busy time, and gaps between consecutive kernels split by what follows
by = collections.defaultdict(list)
for r in win:
    by[r["Stream_Id"]].append(r)
gap_after = collections.defaultdict(list)
dur = collections.defaultdict(list)
for s, rs in by.items():
    for a, b in zip(rs, rs[1:]):
        gap_after[a["n"][:18] + " -> " + b["n"][:18]].append((b["s"] - a["e"]) / 1e3)
    for r in rs:
        dur[r["n"]].append((r["e"] - r["s"]) / 1e3)
print("per-stream busy fraction:", {s: round(sum(r["e"] - r["s"] for r in rs) / (T1 - T0), 2) for s, rs in sorted(by.items())})
print("durations (us): name, calls, mean")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:28s} {len(v):5d} {sum(v) / len(v):8.1f}")
print("gaps between consecutive kernels of one stream (us): transition, count, mean")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {k:42s} {len(v):5d} {sum(v) / len(v):8.1f}")
