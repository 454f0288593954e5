export TMPDIR=/tmp
WARP_AB_ARGS="--members 12" bash tools/ab_warp.sh product prefc
bash tools/ab_bench.sh product prefc
bash tools/ab_bench.sh product prefc
