#!/bin/bash
export TMPDIR=/tmp
for vb in 8 16; do for c in 6 3 2 1; do
  echo "== vb $vb chunk $c"
  HF_EXP_WARP_VB=$vb HF_EXP_WARP_CHUNK=$c python tools/microbench.py --hdr 0 --H 1080 --W 1920 2>&1 | grep -i "fused period (5\|fused period mode 2"
done; done
echo "== HDR 1080p"
for vb in 8 16; do for c in 6 1; do echo "vb $vb chunk $c"; HF_EXP_WARP_VB=$vb HF_EXP_WARP_CHUNK=$c python tools/microbench.py --hdr 1 --H 1080 --W 1920 2>&1 | grep -i "fused period (5\|fused period mode 2"; done; done
