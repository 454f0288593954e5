#!/bin/bash
# A-B of the large-window kernel's workgroup size inside the pipeline (build the variant first: tools/build_variant.sh big4 -DHF_BIG_ONE_WAVE_MIN_BATCH=1000):
# big4 = four-wave workgroups always (rounds 2-5)
for wl in sdr1080_24to60 hdr1080_24to120 sdr1080_64pairs; do echo $wl; AB_ARGS="--workload $wl" bash tools/ab_bench.sh product big4; done
