export TMPDIR=/tmp
python -m pytest tests/test_counters_gpu.py tests/test_sad_reuse_gpu.py tests/test_batch_gpu.py -x -q 2>&1 | tail -6
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs"
run() { python bench.py $Q "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-66s %8.0f frames/s  chain %6.1f us/pair' % ('$*', d['value'], 1e3*(d['ms_per_flow_calc'] or 0)))"; }
for sc in bench chaotic cut; do
run --workload sdr1080_24to60 --steps 16 --scene $sc
run --workload sdr1080_24to60 --steps 16 --scene $sc --no-sad-reuse
run --scene $sc
run --scene $sc --no-sad-reuse
done
run --workload sdr1080_24to60 --steps 16 --pool-order wrap
run --workload sdr1080_24to60 --steps 16 --pool-order wrap --no-sad-reuse
run --pool-order wrap
run --pool-order wrap --no-sad-reuse
