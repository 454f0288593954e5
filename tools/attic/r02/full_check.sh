#!/bin/bash
# full GPU suite + default bench (extra args go to bench.py)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_full; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --no-cpu-baseline --no-reference --no-host-io "$@" > $O/bench.json 2> $O/bench.err
python3 tools/r02/show_bench.py $O/bench.json
