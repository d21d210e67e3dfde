#!/bin/bash
# Scratch: registers, spills, LDS and occupancy of the traversal kernels as the compiler reports them.
# Usage: tools/kernel_resources.sh [extra compiler flags...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -Wno-unused-function \
    -Iinclude -Ibuild "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/kr.o path-tracer_amd/csrc/pt_hip.hip > /tmp/kr.txt 2>&1
python3 - <<'PY'
import re
cur=None; rows={}
for l in open('/tmp/kr.txt'):
    m=re.search(r'remark: (.*) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        cur=t.split(':',1)[1].strip(); rows[cur]={}
    elif cur and ':' in t:
        k,v=t.split(':',1); rows[cur][k.strip()]=v.strip()
for name,r in rows.items():
    if not re.search(r'k_(extend|shadow|generate)_waveINS_7NoCount|k_shadeE|k_(intersect|occluded)_waveINS_7NoCount|k_finishINS_7NoCount', name): continue
    short=re.sub(r'_ZN3ptd\d+','',name)[:48]
    print(f"{short:48s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} spill {r.get('VGPR Spill')} scratch {r.get('ScratchSize [bytes/lane]')} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
PY
