#!/bin/bash
# tools/ab_chain.sh VARIANT... -- stand-alone chain times (batch 1 and 16) of several builds (hopperrender_amd/lib/exp/<v>/)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then L=""; else L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi
  echo "== $v"; HF_LIB=$L python tools/chain_time.py --batch 1 16 2>&1 | grep "flow chain"
  HF_LIB=$L python tools/chain_time.py --batch 1 16 --hdr 0 --H 1080 --W 1920 2>&1 | grep "flow chain"
done
