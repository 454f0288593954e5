#!/bin/bash
# tools/r05/envknobs.sh -- runtime switches nobody had looked at: where kernel arguments live, how graphs are launched, how the host waits
export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
run() {
  echo "== $*"
  env "$@" python tools/chain_time.py --batch 1 16 2>&1 | grep "flow chain"
  env "$@" python bench.py $Q 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %8.0f frames/s  isolated flow calc %.1f us  host %.1f ms' % (d['value'], 1e3*d['ms_per_flow_calc_isolated'], d['host_enqueue_ms_per_step']))"
}
for rep in 1 2; do
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HSA_ENABLE_INTERRUPT=0
done
