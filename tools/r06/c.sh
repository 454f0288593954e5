export TMPDIR=/tmp
AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/ab_bench.sh notab nofresh
bash tools/ab_bench.sh notab nofresh
