export TMPDIR=/tmp
bash tools/ab_bench.sh product reach312 reach184
AB_ARGS="--streams 64 --batch 16" bash tools/ab_bench.sh product reach184
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh product reach184
