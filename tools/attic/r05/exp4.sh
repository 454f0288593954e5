#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05_exp4; mkdir -p $O
Q="--steps 6 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
for v in product no_nb no_bias; do
  HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python bench.py $Q --timeline-out $O/tl_$v.json > $O/b_$v.json 2>$O/err_$v.txt
  HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/chain_time.py --batch 12 16 | sed "s/^/$v /"
  python3 - <<PY
import json
d=json.load(open("$O/tl_$v.json"))
print("$v", " ".join("%s %.1f" % (k, x.get("alone_mean_us",0)) for k,x in sorted(d["kernels"].items())))
PY
done
