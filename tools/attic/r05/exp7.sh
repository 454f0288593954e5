#!/bin/bash
export TMPDIR=/tmp
bash tools/ab_bench.sh product combo combo_ld
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh product combo combo_ld
