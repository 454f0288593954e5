#!/bin/bash
# chain time alone (batch 8, 16) + parity subset for several tools/r02/exp/<name> libraries
export TMPDIR=/tmp; R=$PWD
for v in "$@"; do echo "== $v"
  HF_LIB=$R/tools/r02/exp/$v/libhopperflow.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_batch_gpu.py -x -q -m gpu 2>&1 | tail -1
  HF_LIB=$R/tools/r02/exp/$v/libhopperflow.so python tools/chain_time.py --batch 8 16 2>&1 | tail -2
  HF_LIB=$R/tools/r02/exp/$v/libhopperflow.so python tools/chain_time.py --batch 16 --hdr 0 --H 1080 --W 1920 2>&1 | tail -1
done
