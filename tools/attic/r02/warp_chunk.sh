#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_batch_period_gpu.py tests/test_parity_gpu.py tests/test_random_gpu.py -x -q -m gpu 2>&1 | tail -3
for c in ${CHUNKS:-6 1}; do
  echo "== chunk $c"
  HF_EXP_WARP_CHUNK=$c python tools/microbench.py 2>&1 | grep -i "fused\|warp mode 2 real"
  HF_EXP_WARP_CHUNK=$c python tools/microbench.py --hdr 0 --H 1080 --W 1920 2>&1 | grep -i "fused\|warp mode 2 real"
done
