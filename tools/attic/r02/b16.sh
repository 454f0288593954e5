#!/bin/bash
export TMPDIR=/tmp
python tools/chain_time.py --batch 8 12 16 2>&1 | tail -3
python tools/chain_time.py --batch 8 16 --hdr 0 --H 1080 --W 1920 2>&1 | tail -2
printf '16 8\n32 16\n16 16\n24 12\n' > /tmp/s.txt; bash tools/r02/scan_op.sh < /tmp/s.txt
printf '24 8\n32 16\n48 16\n' > /tmp/s.txt; WORKLOAD=sdr1080_24to60 bash tools/r02/scan_op.sh < /tmp/s.txt
