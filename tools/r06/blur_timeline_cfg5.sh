for v in product blurold; do if [ $v = product ]; then L=""; else L=$PWD/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi
HF_LIB=$L python bench.py --workload hdr2160_nb10_blur32 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs --steps 8 --warmup 2 --timeline-out gpurun_out/tl5_$v.json 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=json.load(open('gpurun_out/tl5_$v.json'))
print('$v', round(d['value']), 'period', t['mean_period_ms_per_queue'], 'alone', t['alone_kernel_time_per_batch_period_us'])
for k,v in t['kernels'].items():
    if k in ('blur','warp_period','level_2'): print('    %-18s n=%4d mean %8.1f alone %s' % (k, v['n'], v['mean_us'], v.get('alone_mean_us')))
"
done
