#!/bin/bash
# per-kernel stats of chain_time.py for an experimental library: prof_variant.sh <variant> <batch>
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_var; rm -rf $O; mkdir -p $O
cd /tmp; HF_LIB=$R/tools/r02/exp/$1/libhopperflow.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/tools/chain_time.py --batch $2 --n 50 > /dev/null 2>&1; cd $R
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/p_kernel_stats.csv")):
    print("  %-60s calls %5s avg %9.1f us" % (r["Name"][28:88], r["Calls"], float(r["AverageNs"])/1e3))
PY
