#!/bin/bash
# full GPU suite + default bench
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_full; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 100 --no-cpu-baseline --no-reference "$@" > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
print("value", j["value"], "ms_per_step", j["ms_per_step"], "host_enqueue", j["host_enqueue_ms_per_step"], "flow", j["ms_per_flow_calc"], j["ms_per_flow_calc_isolated"], "warp_us", j["roofline"]["avg_launch_us"], j["roofline"]["isolated"]["avg_launch_us"])
PY
