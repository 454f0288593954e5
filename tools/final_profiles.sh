#!/bin/bash
# tools/final_profiles.sh TAG -- the measurements the round's profiles/ are made from (run on the GPU box, then
# `python tools/copy_profiles.py TAG` in the build container):
#   default bench lines (HDR 2160p, SDR 1080p, BASELINE configs 4 and 5), rocprofv3 kernel stats of the same commands and of one
#   stream alone, PMC FETCH_SIZE / WRITE_SIZE passes of the fused period warp alone (one launch at a time) AND of the whole
#   batched pipeline at the bench's operating point, stand-alone kernel times.
#   `tools/final_profiles.sh TAG bench` re-runs only the four bench lines -- after `copy_profiles.py` has regenerated
#   profiles/roofline_traffic.json from the PMC passes, so that the committed lines price their bytes with THIS round's traffic record.
TAG=${1:-r06}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG
if [ "$2" != bench ]; then rm -rf $O; fi
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python bench.py --workload sdr1080_24to60 --no-reference > $O/bench_sdr1080.json 2> $O/bench_sdr1080.err
python bench.py --workload sdr1080_64pairs --no-reference --no-cpu-baseline --no-host-io > $O/bench_sdr1080_64pairs.json 2> $O/bench_cfg4.err
python bench.py --workload hdr2160_nb10_blur32 --no-reference --no-cpu-baseline --no-host-io > $O/bench_hdr2160_nb10_blur32.json 2> $O/bench_cfg5.err
python bench.py --workload hdr1080_24to120 --no-reference --no-cpu-baseline --no-host-io > $O/bench_hdr1080.json 2> $O/bench_hdr1080.err
python bench.py --workload sdr2160_24to60 --no-reference --no-cpu-baseline --no-host-io > $O/bench_sdr2160.json 2> $O/bench_sdr2160.err
python bench.py --pool-order wrap --no-reference --no-cpu-baseline --no-host-io --no-other-workloads --no-content-legs > $O/bench_default_wrap6.json 2>> $O/bench_default.err
python bench.py --no-sad-reuse --no-reference --no-cpu-baseline --no-host-io --no-other-workloads --no-content-legs > $O/bench_default_no_sad_reuse.json 2>> $O/bench_default.err
python bench.py --workload sdr1080_24to60 --no-sad-reuse --no-reference --no-cpu-baseline --no-host-io > $O/bench_sdr1080_no_sad_reuse.json 2>> $O/bench_sdr1080.err
if [ "$2" = bench ]; then ls $O; exit 0; fi
# the pipeline's timeline WITHOUT a profiler (hf_batch_timeline_*: start / stop events of every dispatch, one step in the middle of the timed
# region, plus a stand-alone leg of one batch), bracketed by the plain lines above / below: concurrency per kernel, stretch vs stand-alone,
# idle gaps per queue (tools/timeline_report.py)
TQ="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs"
python bench.py $TQ --timeline-out $O/pipeline_timeline.json > $O/bench_default_timeline_run.json 2>> $O/bench_default.err
python bench.py $TQ --workload sdr1080_24to60 --timeline-out $O/pipeline_timeline_sdr1080.json > $O/bench_sdr1080_timeline_run.json 2>> $O/bench_sdr1080.err
python bench.py $TQ > $O/bench_default_plain_after_timeline.json 2>> $O/bench_default.err
python tools/microbench.py > $O/microbench.txt 2>&1
python tools/microbench.py --hdr 0 --H 1080 --W 1920 >> $O/microbench.txt 2>&1
python tools/chain_time.py --batch 1 2 4 8 16 >> $O/microbench.txt 2>&1
python tools/chain_time.py --batch 1 8 16 --hdr 0 --H 1080 --W 1920 >> $O/microbench.txt 2>&1
for sc in static pan64 chaotic cut; do python tools/chain_time.py --batch 1 16 --scene $sc >> $O/microbench.txt 2>&1; done
python tools/chain_time.py --batch 1 12 16 --no-reuse >> $O/microbench.txt 2>&1
python tools/warp_ab.py >> $O/microbench.txt 2>&1
cd /tmp
Q="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python3 $R/bench.py $Q > $O/bench_default_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sdr1080 -o p -- python3 $R/bench.py --workload sdr1080_24to60 $Q > $O/bench_sdr1080_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streams1 -o p -- python3 $R/bench.py --streams 1 --batch 1 --steps 2 --periods-per-step 20 $Q > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain16 -o p -- python3 $R/tools/chain_time.py --batch 16 --n 50 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain16_static -o p -- python3 $R/tools/chain_time.py --batch 16 --n 50 --scene static > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain16_chaotic -o p -- python3 $R/tools/chain_time.py --batch 16 --n 50 --scene chaotic > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain16_noreuse -o p -- python3 $R/tools/chain_time.py --batch 16 --n 50 --no-reuse > /dev/null 2>&1
for wl in hdr1080_24to120 sdr2160_24to60; do for c in FETCH_SIZE WRITE_SIZE; do   # the pipeline's bytes of the two workloads added in round 6
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcpipe_${wl}_$c -o p -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --periods-per-step 8 --no-profile $Q > /dev/null 2>&1
  echo "pmc pipeline $wl $c rc=$?"
done; done
for wl in hdr2160_24to120 sdr1080_24to60; do for c in FETCH_SIZE WRITE_SIZE; do
  # the fused period warp alone: one member, one launch at a time
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${wl}_$c -o p -- python3 $R/bench.py --workload $wl --streams 1 --batch 1 --steps 2 --warmup 1 --periods-per-step 12 --no-profile $Q > /dev/null 2>&1
  echo "pmc warp $wl $c rc=$?"
  # the whole pipeline at the bench's operating point (2 batch streams of 16; counter collection serialises the dispatches)
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcpipe_${wl}_$c -o p -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --periods-per-step 8 --no-profile $Q > /dev/null 2>&1
  echo "pmc pipeline $wl $c rc=$?"
done; done
cd $R
# instructions per wave of the staged period warp (16-member launches) and its SQ / TD / TA / TCP / TCC counters
bash tools/pmc_warp_valu.sh product > $O/pmc_warp_valu.txt 2>&1
bash tools/pmc_warp_batch.sh 16 > $O/pmc_warp_wg_kernel.txt 2>&1
# the 12 chain launches alone, 16 pairs per launch: busy fractions, occupancy, L2 -> L1 bytes
{ echo "# tools/pmc_chain_batch.sh 16: the 12 launches of the flow chain alone, 16 pairs per launch (2160p HDR = 480 x 270 grid), read by tools/pmc_chain_derive.py."
  echo "# Useful candidate bytes per launch (16 pairs x 129,600 pixels x 16 candidates x 4 B) = 132.7 MB per axis: 1 axis for flow_big_partial, 2 for the level kernels."
  echo "# valu% = vector ALU busy, wait% = share of wave-cycles spent waiting, td% / tcc% = texture-data unit / L2 channels busy, L2->L1 = TCP_TCC_READ_REQ x 64 B."
  bash tools/pmc_chain_batch.sh 16 2>&1 | grep -v " rc=0$"; } > $O/pmc_chain_batch16.txt
python3 tools/chain_beside_warp.py $O/stats_default/p_kernel_trace.csv > $O/chain_beside_warp.txt 2>&1
# the library's event timeline and rocprofv3's kernel trace of the SAME dispatches, one process: what a start event measures, and what the
# profiler does to the pipeline (tools/timeline_vs_trace.py)
cd /tmp
for wl in hdr2160_24to120 sdr1080_24to60; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tlx_$wl -o p -- python3 $R/bench.py $Q --steps 6 --warmup 2 --workload $wl --timeline-out $O/tlx_$wl.json > $O/tlx_bench_$wl.json 2> /dev/null
  { python3 $R/tools/timeline_vs_trace.py $O/tlx_${wl}_raw.json $O/tlx_$wl/p_kernel_trace.csv
    python3 -c "
import json,sys
t=json.load(open('$O/tlx_$wl.json')); b=json.loads(open('$O/tlx_bench_$wl.json').read().strip().splitlines()[-1])
print('# this (profiled) run: %.0f frames/s, host enqueue %.1f ms of a %.1f ms step; queues busy by the event view: %s; mean queues busy %.2f' % (b['value'], b['host_enqueue_ms_per_step'], b['ms_per_step'], [q['busy_frac'] for q in t['queues']], t['concurrency']['mean_queues_busy']))"
  } > $O/timeline_vs_trace_$wl.txt 2>&1
  rm -rf $O/tlx_$wl $O/tlx_${wl}_raw.json
done
cd $R
ls $O
