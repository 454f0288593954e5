export TMPDIR=/tmp
python -m pytest tests/test_lds_rows_gpu.py tests/test_parity_gpu.py tests/test_random_gpu.py tests/test_batch_gpu.py tests/test_debug_bounds_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for v in before product; do
HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/chain_time.py --batch 12 16 | sed "s/^/$v /"
HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/chain_time.py --batch 12 --hdr 0 --H 1080 --W 1920 | sed "s/^/$v /"
done; done
bash tools/ab_bench.sh before product
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh before product
