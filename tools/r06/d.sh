export TMPDIR=/tmp
python -m pytest tests/test_counters_gpu.py tests/test_timeline_gpu.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -15
( time python bench.py --no-cpu-baseline --no-reference --no-host-io > gpurun_out/bench_r6_dev.json 2> gpurun_out/bench_r6_dev.err ) 2>&1 | tail -3
tail -c 1500 gpurun_out/bench_r6_dev.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r6_dev.json') if l.startswith('{')][-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'commit', d['roofline'].get('traffic_record_commit'))
for k,v in d.get('other_workloads',{}).items(): print(' other', k, v if not isinstance(v,dict) else {x:v[x] for x in ('value','ms_per_flow_calc') if x in v} or v)
for wl,legs in d.get('content',{}).items():
    if isinstance(legs,dict):
        for sc,v in legs.items(): print(' content', wl, sc, json.dumps(v)[:420])
    else: print(' content', wl, legs)
PY
