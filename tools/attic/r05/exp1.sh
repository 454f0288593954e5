#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05_exp1; mkdir -p $O
python -m pytest tests/test_timeline_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab_bench.sh product nt_a nt_ab
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh product
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
python bench.py $Q --streams 52 --batch 13 --timeline-out $O/tl_4x13.json > $O/b_4x13.json 2>$O/err.txt
python bench.py $Q --streams 64 --batch 16 --timeline-out $O/tl_4x16.json > $O/b_4x16.json 2>>$O/err.txt
python3 - <<'PY'
import json
for n in ("4x13","4x16"):
    try:
        b=json.loads(open(f"gpurun_out/r05_exp1/b_{n}.json").read().strip().splitlines()[-1]); d=json.load(open(f"gpurun_out/r05_exp1/tl_{n}.json"))
        print(n, b["value"], "frames/s; period", d["mean_period_ms_per_queue"], "alone sum", d.get("alone_kernel_time_per_batch_period_us"))
        for k,v in d["kernels"].items(): print("   %-18s mean %8.1f alone %8.1f stretch %5.2f" % (k, v["mean_us"], v.get("alone_mean_us",0), v.get("stretch_vs_alone",0)))
    except Exception as e: print(n, "failed", e)
PY
