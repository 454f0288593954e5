#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py tests/test_deferred_planes_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_bench.sh product st3 a18 a3 blur_nt
