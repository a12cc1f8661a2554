#!/bin/bash
# per-kernel register / scratch usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: tools/kres.sh <file.hip> [name filter] [extra flags]
F=$1; PAT=${2:-.}; shift; shift
EX=""; [ "$(basename $F)" = "vil_attn_mfma.hip" ] && EX="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $EX "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null $F 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=[m.group(1)]; rows.append(cur); continue
    m=re.search(r'(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)',l)
    if m and cur is not None: cur.append(m.group(1).split()[0][:7]+'='+m.group(2))
for r in rows:
    if re.search(r'$PAT', r[0]): print(r[0][:70].ljust(70), ' '.join(r[1:]))
"
