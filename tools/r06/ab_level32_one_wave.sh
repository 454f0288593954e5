#!/bin/bash
# A-B inside the pipeline of level 32 as one wave per window (product, batches of 4 and more) against the four-wave workgroup
# (tools/build_variant.sh l32w4 -DHF_LEVEL32_ONE_WAVE_MIN_BATCH=1000), then the chain alone
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_sad_reuse_gpu.py tests/test_counters_gpu.py tests/test_batch_period_gpu.py -x -q 2>&1 | tail -3
for wl in sdr1080_24to60 sdr1080_64pairs hdr1080_24to120 hdr2160_24to120 sdr2160_24to60; do echo $wl; AB_ARGS="--workload $wl" bash tools/ab_bench.sh product l32w4; done
for v in product l32w4; do if [ $v = product ]; then L=""; else L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi; echo $v; HF_LIB=$L python tools/chain_time.py --batch 4 12 16 2>&1 | tail -3; done
