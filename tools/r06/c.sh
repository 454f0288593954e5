export TMPDIR=/tmp
python -m pytest tests/test_sad_reuse_gpu.py tests/test_lds_rows_gpu.py -x -q 2>&1 | tail -3
for v in product g4; do echo "== $v"; L=""; [ $v != product ] && L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; HF_LIB=$L bash tools/chain_stats.sh 16 2>&1 | grep -v rocprofv3 | grep small_kernel; HF_LIB=$L python tools/chain_time.py --batch 1 12 16 | grep "flow chain"; done
AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/ab_bench.sh fused product g4
