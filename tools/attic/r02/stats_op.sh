#!/bin/bash
# rocprofv3 kernel stats of one operating point: stats_op.sh <workload> <streams> <batch>
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_stats_op; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o p -- python3 $R/bench.py --workload $1 --streams $2 --batch $3 --steps 6 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io > $O/b.json 2>/dev/null
cd $R; python3 tools/r02/show_bench.py $O/b.json
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/st/p_kernel_stats.csv")):
    print("  %-64s calls %6s avg %9.1f us  %6s %%" % (r["Name"][28:92], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
