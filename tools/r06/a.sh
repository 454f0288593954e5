export TMPDIR=/tmp
python -m pytest tests/test_sad_reuse_gpu.py -x -q 2>&1 | tail -15
for sc in bench chaotic; do
python tools/chain_time.py --batch 1 16 --scene $sc 2>&1 | grep "flow chain"
python tools/chain_time.py --batch 1 16 --scene $sc --no-reuse 2>&1 | grep "flow chain"
python tools/chain_time.py --batch 16 --hdr 0 --H 1080 --W 1920 --scene $sc 2>&1 | grep "flow chain"
python tools/chain_time.py --batch 16 --hdr 0 --H 1080 --W 1920 --scene $sc --no-reuse 2>&1 | grep "flow chain"
done
