#!/bin/bash
# Register / spill / occupancy summary of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage):
#   bash tools/kernel_resources.sh rag_amd/csrc/conv3d_x3.hip [name filter regex]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -Iinclude -Irag_amd/csrc -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  python3 -c "
import re,sys,subprocess
cur={}
rows=[]
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?): (.*?) \[-Rpass', line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=='Function Name':
        if cur: rows.append(cur)
        cur={'name':v}
    else: cur[k]=v
if cur: rows.append(cur)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=n.replace('ragmi::','').replace('(K3Args, X3Extra)','').replace('void ','')
    print(f\"{n[:64]:64s} VGPR {r.get('VGPRs','?'):>3s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s}\")
" | grep -E "$filt"
