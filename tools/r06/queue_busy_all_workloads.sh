for wl in hdr1080_24to120 sdr2160_24to60 sdr1080_64pairs hdr2160_nb10_blur32 sdr360_24to60; do
python bench.py --workload $wl --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs --steps 8 --warmup 2 --timeline-out gpurun_out/tlq_$wl.json 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=json.load(open('gpurun_out/tlq_$wl.json'))
print('$wl', round(d['value']), 'ms/step', d['ms_per_step'], 'host_enqueue', d['host_enqueue_ms_per_step'], 'busy', [q['busy_frac'] for q in t['queues']], 'queues busy', t['concurrency']['mean_queues_busy'], 'period', t['mean_period_ms_per_queue'], 'alone', t['alone_kernel_time_per_batch_period_us'])
for k,v in t['kernels'].items(): print('    %-18s n=%4d mean %8.1f alone %s' % (k, v['n'], v['mean_us'], v.get('alone_mean_us')))
"
done
