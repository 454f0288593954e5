export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
HF_LIB=$PWD/hopperrender_amd/lib/libhopperflow_dbg.so python -m pytest tests -m gpu -x -q 2>&1 | tail -4
( time python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_full.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic_record_commit'), list(d.keys())[-8:])
PY
