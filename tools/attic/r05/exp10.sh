export TMPDIR=/tmp
for rep in 1 2; do for v in product single_nt; do
export HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so
python tools/microbench.py 2>&1 | grep -i "flow chain\|updateFrame" | sed "s/^/$v /"
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v isolated flow', d['ms_per_flow_calc_isolated'], d['ms_per_flow_calc_isolated_after_sync'])"
done; done
