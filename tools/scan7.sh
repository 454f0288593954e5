#!/bin/bash
for lds in 0 32768 40960 53248 65536; do
  echo "lds=$lds $(HF_WARP_LDS=$lds python tools/microbench.py --n 20 2>&1 | grep -E 'fused period mode 2, HBM-cold' | awk '{print "fused cold", $6, "us"}') $(HF_WARP_LDS=$lds python bench.py --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench frames/s', j['value'])")"
done
