#!/bin/bash
export TMPDIR=/tmp
WARP_AB_ARGS="--members 12" bash tools/ab_warp.sh product tpw2 tpw2_skipa tpw3_skipa
bash tools/ab_bench.sh product tpw2 tpw2_skipa tpw3_skipa
