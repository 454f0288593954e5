export TMPDIR=/tmp
python -m pytest tests/test_lds_rows_gpu.py -x -q 2>&1 | tail -2
bash tools/ab_bench.sh product nolds32 noldsbig
AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/ab_bench.sh product nolds32 noldsbig
