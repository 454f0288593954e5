#!/bin/bash
# the library's event timeline and rocprofv3's kernel trace of the same dispatches, one process (tools/timeline_vs_trace.py)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r06_tlx; rm -rf $O; mkdir -p $O
Q="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs --steps 6 --warmup 2"
cd /tmp
for wl in hdr2160_24to120 sdr1080_24to60; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$wl -o p -- python3 $R/bench.py $Q --workload $wl --timeline-out $O/tl_$wl.json > $O/bench_$wl.json 2> $O/err_$wl.txt
  python3 $R/tools/timeline_vs_trace.py $O/tl_${wl}_raw.json $O/tr_$wl/p_kernel_trace.csv > $O/timeline_vs_trace_$wl.txt 2>&1
  cat $O/timeline_vs_trace_$wl.txt
  rm -rf $O/tr_$wl
done
