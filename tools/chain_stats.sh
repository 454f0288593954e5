#!/bin/bash
# tools/chain_stats.sh [BATCH [chain_time.py options]] -- rocprofv3 --kernel-trace --stats of the batched flow chain alone (tools/chain_time.py): per-kernel
# average durations of the 12 launches, written to gpurun_out/chain_stats/ (copied to profiles/ by tools/copy_profiles.py)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/chain_stats; rm -rf $O; mkdir -p $O; B=${1:-16}; shift
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chain -- python3 $R/tools/chain_time.py --batch $B --n 50 "$@" > $O/chain.log 2>&1
cd $R; tail -2 $O/chain.log; python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    n=r["Name"]
    if "hf::" in n and ("flow_" in n or "blur" in n):
        k=n.replace("hf::(anonymous namespace)::","").replace("void ","").split("(")[0]
        print("%-48s calls %4d avg %8.2f us" % (k, int(r["Calls"]), float(r["AverageNs"])/1e3)); 
PY
