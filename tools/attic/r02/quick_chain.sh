#!/bin/bash
# parity subset + chain timing + per-kernel stats (batch 8)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_quick; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_batch_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/chain_time.py --batch 1 2 8 2>&1 | tail -3
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain8 -o p -- python3 $R/tools/chain_time.py --batch 8 --n 50 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/stats_chain8/p_kernel_stats.csv")):
    print("  %-60s calls %5s avg %9.1f ns" % (r["Name"][28:88], r["Calls"], float(r["AverageNs"])))
PY
