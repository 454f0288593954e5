export TMPDIR=/tmp
python -m pytest tests/test_sad_reuse_gpu.py tests/test_lds_rows_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -3
for v in notab fused; do echo "== $v"; HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so bash tools/chain_stats.sh 16 2>&1 | grep -v rocprofv3; done
AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/ab_bench.sh notab fused
bash tools/ab_bench.sh notab fused
