#!/bin/bash
# tools/scan_op.sh "STREAMS BATCH" ... -- the default workload at several operating points (pair streams per GPU, pairs per batch), one box
export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs ${AB_ARGS}"
for rep in 1 2; do for sb in "$@"; do read S B <<< "$sb"
  python bench.py $Q --streams $S --batch $B 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%3s x %2s  %8.0f frames/s  ms/step %7.2f  warp-in-pipe %7.1f us  chain %6.1f us/pair' % ('$S', '$B', d['value'], d['ms_per_step'], r.get('kernel_in_pipeline',{}).get('avg_launch_us',0), 1e3*(d['ms_per_flow_calc'] or 0)))"
done; done
