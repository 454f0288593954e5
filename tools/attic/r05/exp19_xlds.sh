#!/bin/bash
# tools/r05/xlds.sh -- X step of the uniform-offset tiles out of LDS: parity (product + debug-bounds build), then chain alone and pipeline,
# same box, against builds without it (x0) and with one kernel each (x1 = windows > 32, x2 = level 32, x4 = level 16).
export TMPDIR=/tmp
O=gpurun_out/r05_xlds; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_lds_rows_gpu.py tests/test_random_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
HF_LIB=$PWD/hopperrender_amd/lib/libhopperflow_dbg.so timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_debug_bounds_gpu.py tests/test_lds_rows_gpu.py -x -q -m gpu > $O/dbg.txt 2>&1; tail -3 $O/dbg.txt
bash tools/ab_chain.sh x0 product x1 x2 x4 x0 product | tee $O/chain.txt
bash tools/ab_bench.sh x0 product | tee $O/ab_hdr2160.txt
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh x0 product | tee $O/ab_sdr1080.txt
