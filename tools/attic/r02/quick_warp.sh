#!/bin/bash
# parity of the warp paths + stand-alone timings of the fused period warp
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_warp; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py tests/test_batch_period_gpu.py tests/test_properties_gpu.py -x -q -m gpu 2>&1 | tail -6
python tools/microbench.py 2>&1 | grep -i "fused\|warp mode 2 real"
python tools/microbench.py --hdr 0 --H 1080 --W 1920 2>&1 | grep -i "fused\|warp mode 2 real"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/tools/microbench.py --n 20 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/stats/p_kernel_stats.csv")):
    if "warp" in r["Name"] or "disp" in r["Name"]: print("  %-90s calls %5s avg %9.1f ns" % (r["Name"][28:118], r["Calls"], float(r["AverageNs"])))
PY
