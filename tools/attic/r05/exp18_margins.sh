#!/bin/bash
# tools/r05/margins.sh -- reduced plane margins: the new tests + the parity suites (product and debug-bounds builds), then a same-box A/B
# against the build with every reachable offset in the margins (hopperrender_amd/lib/exp/before).
export TMPDIR=/tmp
O=gpurun_out/r05_margins; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_margins_gpu.py -x -q -m gpu > $O/new_tests.txt 2>&1; tail -5 $O/new_tests.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_lds_rows_gpu.py tests/test_random_gpu.py tests/test_batch_gpu.py tests/test_deferred_planes_gpu.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
HF_LIB=$PWD/hopperrender_amd/lib/libhopperflow_dbg.so timeout 1500 python -m pytest tests/test_plane_margins_gpu.py tests/test_debug_bounds_gpu.py tests/test_lds_rows_gpu.py -x -q -m gpu > $O/dbg.txt 2>&1; tail -3 $O/dbg.txt
for v in before product; do
  if [ "$v" = product ]; then L=""; else L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi
  echo "== chain alone, $v"; HF_LIB=$L python tools/chain_time.py --batch 1 12 2>&1 | tail -2
  HF_LIB=$L python tools/chain_time.py --hdr 0 --H 1080 --W 1920 --batch 1 12 2>&1 | tail -2
done | tee $O/chain.txt
bash tools/ab_bench.sh before product | tee $O/ab_hdr2160.txt
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh before product | tee $O/ab_sdr1080.txt
