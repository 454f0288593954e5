#!/bin/bash
# tools/ab_bench.sh VARIANT... -- the default bench (short) for several builds of the library (hopperrender_amd/lib/exp/<v>/, see
# tools/build_variant.sh), alternating, on ONE box: A-B of kernel variants inside the pipeline.  "product" = the in-tree build.
export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs ${AB_ARGS}"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then L=""; else L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi
  HF_LIB=$L python bench.py $Q 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %8.0f frames/s  ms/step %7.2f  warp-in-pipe %7.1f us  chain %6.1f us/pair  host %5.1f ms' % ('$v', d['value'], d['ms_per_step'], r.get('kernel_in_pipeline',{}).get('avg_launch_us',0), 1e3*(d['ms_per_flow_calc'] or 0), d['host_enqueue_ms_per_step']))"
done; done
