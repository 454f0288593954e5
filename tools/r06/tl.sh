export TMPDIR=/tmp
Q="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --workload sdr1080_24to60 --steps 16 --warmup 2"
for v in notab fused nofresh; do
HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python bench.py $Q --timeline-out gpurun_out/tl_$v.json 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d.get('timeline'))"
python3 - <<PY
import json
r=json.load(open('gpurun_out/tl_$v.json'))
ks=r.get('kernels') or r.get('per_kernel') or {}
print(list(r.keys()))
for k,v in sorted(ks.items()):
    print('%-22s'%k, {kk: vv for kk,vv in v.items() if kk in ('n','mean_us','alone_mean_us','stretch','share_of_busy')})
PY
done
