#!/bin/bash
for s in 4 2 1; do for u in 1 2 4; do
  echo "split=$s upw=$u: $(HF_WARP_SPLIT=$s HF_WARP_UPW=$u python tools/microbench.py --n 30 2>&1 | grep -E 'warp mode 2 real|fused' | awk '{printf "%s %s us | ", $1=="fused"?"fused":"single", ($1=="fused")?$7:$7}')"
done; done
