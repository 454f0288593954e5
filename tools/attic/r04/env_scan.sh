export TMPDIR=/tmp
Q="--steps 6 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-profile"
run() { echo -n "$1: "; env $1 python bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])"; }
for rep in 1 2; do
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HSA_ENABLE_SDMA=0"
run "GPU_MAX_HW_QUEUES=4"
run "AMD_DIRECT_DISPATCH=0"
done
