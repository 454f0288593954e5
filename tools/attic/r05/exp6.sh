#!/bin/bash
export TMPDIR=/tmp
bash tools/ab_bench.sh product plane_nt yst_nt x_nt grid_nt
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh product yst_nt x_nt
