#!/bin/bash
# stopwatch builds of the fused warp (tools/r02/exp/wi*: WRONG results, timing only): fused period alone + 16-member warp-only pipeline
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_wi; mkdir -p $O
for v in "$@"; do
  L=$R/tools/r02/exp/$v/libhopperflow.so
  echo "== $v"
  HF_LIB=$L python tools/microbench.py 2>&1 | grep -i "fused" | head -3
  HF_LIB=$L python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io --streams 16 --batch 16 --diagnose no-flow > $O/$v.json 2> $O/$v.err
  python3 tools/r02/show_bench.py $O/$v.json
done
