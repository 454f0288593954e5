#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel inside the default pipeline (cold caches, two batch streams)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_pipe; rm -rf $O; mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --periods-per-step 8 --no-profile --no-cpu-baseline --no-reference --no-host-io "$@" > $O/$c.json 2>/dev/null
  echo "$c rc=$?"
done
cd $R
python3 - <<PY
import csv, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % c)):
        if r["Counter_Name"] == c: acc[r["Kernel_Name"][28:80]].append(float(r["Counter_Value"]))
    tot[c] = acc
names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]))
print("%-54s %8s %12s %12s" % ("kernel", "calls", "fetch MB x2", "write MB"))
for n in names:
    f, w = tot["FETCH_SIZE"].get(n, [0]), tot["WRITE_SIZE"].get(n, [0])
    print("%-54s %8d %12.1f %12.1f" % (n, len(f), 2 * sum(f) / len(f) / 1024, sum(w) / len(w) / 1024))
PY
