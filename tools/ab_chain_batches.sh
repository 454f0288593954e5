export TMPDIR=/tmp
for v in "$@"; do echo "== $v"; HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/chain_time.py --batch 1 2 4 8 --n 60 | sed 's/flow chain 3840x2160 hdr=1 R=16//'; done
