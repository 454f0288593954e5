#!/bin/bash
# full pipeline bench for several tools/r02/exp/<name> libraries on ONE box; extra args after -- go to bench.py
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pv; mkdir -p $O
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=($1); shift; done; shift
for rep in 1 2; do for v in "${libs[@]}"; do
  HF_LIB=$R/tools/r02/exp/$v/libhopperflow.so python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/$v.json 2> $O/$v.err
  echo -n "[$v] "; python3 tools/r02/show_bench.py $O/$v.json
done; done
