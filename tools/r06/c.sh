export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py tests/test_batch_1080p_shapes_gpu.py tests/test_fused_fullsize_gpu.py tests/test_staged_ragged_gpu.py -x -q 2>&1 | tail -3
AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/ab_bench.sh before product
AB_ARGS="--workload sdr2160_24to60" bash tools/ab_bench.sh before product
