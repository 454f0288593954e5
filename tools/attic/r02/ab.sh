#!/bin/bash
# A-B on ONE box: the working tree's library vs tools/r02/exp/prev (alternating runs); extra args go to bench.py
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_ab; mkdir -p $O
for i in 1 2 3; do
  HF_LIB=$R/tools/r02/exp/prev/libhopperflow.so python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/a.json 2> $O/a.err
  echo -n "[prev] "; python3 tools/r02/show_bench.py $O/a.json
  python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/b.json 2> $O/b.err
  echo -n "[new ] "; python3 tools/r02/show_bench.py $O/b.json
done
