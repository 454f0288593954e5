export TMPDIR=/tmp
for rep in 1 2; do for v in product wpe_l4 wpe_l2 wpe_row wpe_all; do
HF_LIB=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so python tools/chain_time.py --batch 12 16 | sed "s/^/$v /"
done; done
bash tools/ab_bench.sh product wpe_l4 wpe_l2 wpe_row wpe_all
AB_ARGS="--workload sdr1080_24to60" bash tools/ab_bench.sh product wpe_l4 wpe_all
