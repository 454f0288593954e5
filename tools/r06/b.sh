export TMPDIR=/tmp
python -m pytest tests/test_sad_reuse_gpu.py -x -q 2>&1 | tail -4
echo "== bench"; bash tools/chain_stats.sh 16 2>&1 | grep -v rocprofv3 | grep v2
echo "== chaotic"; bash tools/chain_stats.sh 16 --scene chaotic 2>&1 | grep -v rocprofv3 | grep v2
