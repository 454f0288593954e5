#!/bin/bash
# tools/kernel_resources.sh FILE.hip [pattern] -- registers / scratch / LDS / occupancy of the gfx950 kernels of one translation unit
# (device-only compile with the product flags + -Rpass-analysis=kernel-resource-usage)
F=${1:-hf_flow.hip}; P=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include --offload-device-only -c hopperrender_amd/csrc/$F -o /tmp/kr_$$.co \
  -Rpass-analysis=kernel-resource-usage $HF_CXXFLAGS 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['c++filt'],input='\n'.join(r['name'] for r in rows),capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=n.replace('hf::(anonymous namespace)::','')
    if re.search(sys.argv[1],n): print('%4s vgpr %3s sgpr %5s scratch %6s lds %2s waves/SIMD  %s'%(r.get('vgpr'),r.get('sgpr'),r.get('scratch'),r.get('lds'),r.get('occ'),n[:100]))
" "$P"; rm -f /tmp/kr_$$.co
