#!/bin/bash
# rocprofv3 kernel stats of a bench configuration: args = bench flags
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_profbench; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 4 --warmup 1 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/bench.json 2> $O/bench.err
cd $R; python3 tools/r02/show_bench.py $O/bench.json
python3 - <<PY
import csv
tot=0; rows=[]
for r in csv.DictReader(open("$O/stats/p_kernel_stats.csv")):
    rows.append(r); tot+=float(r["TotalDurationNs"])
for r in rows:
    print("  %-86s calls %6s avg %9.1f us  %5.1f%%" % (r["Name"][28:114], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
