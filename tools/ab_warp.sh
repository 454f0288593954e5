#!/bin/bash
# tools/ab_warp.sh VARIANT... -- tools/warp_ab.py (fused 2160p HDR period alone: hot / HBM-cold / per member of a 16-member launch + SHA of the outputs)
# for several builds of the library under hopperrender_amd/lib/exp/<variant>/ (tools/build_variant.sh), on ONE box
export TMPDIR=/tmp
for v in "$@"; do HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/warp_ab.py $WARP_AB_ARGS 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-8s hot %6.2f cold %6.2f batch/member %6.2f  sha %s %s' % ('$v', d['single_hot_us'], d['single_cold_us'], d['batch_us_per_member'], d['sha_single'], d['sha_batch']))"; done
