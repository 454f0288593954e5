export TMPDIR=/tmp
echo "== 1080p"; AB_ARGS="--workload sdr1080_24to60 --steps 16" bash tools/scan_op.sh "36 12" "48 12" "48 16" "32 16" "40 10" "32 8"
echo "== 2160p"; bash tools/scan_op.sh "48 12" "64 16" "48 16" "36 12" "40 10"
