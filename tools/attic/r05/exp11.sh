export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-clock-probe"
for rep in 1 2; do for k in 0 1 2 4; do GPU_MAX_HW_QUEUES=${HQ:-5} python bench.py $Q --normal-priority-batches $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('normal-priority batches $k  %8.0f frames/s' % d['value'])"; done; done
for k in 2 4; do GPU_MAX_HW_QUEUES=8 python bench.py $Q --normal-priority-batches $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq 8, normal-priority batches $k  %8.0f frames/s' % d['value'])"; done
