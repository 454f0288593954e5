#!/bin/bash
# tools/r06/hip_api_stats.sh WORKLOAD -- which HIP calls the issuing thread spends its time in (rocprofv3 --hip-trace --stats over a short bench run)
export TMPDIR=/tmp; R=$PWD; WL=${1:-sdr360_24to60}; O=$R/gpurun_out/hipapi_$WL; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs > $O/bench.json 2>/dev/null
head -25 $O/p_hip_api_stats.csv | cut -c1-160
rm -f $O/p_hip_api_trace.csv
