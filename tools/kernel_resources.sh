#!/bin/bash
# VGPRs / spills / occupancy / LDS of every kernel in one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
#   bash tools/kernel_resources.sh sd-webui-text2video_amd/csrc/norm.hip [filter] [extra hipcc flags...]
src=$1; filt=${2:-.}; shift 2 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re, sys
cur = None; rows = []
for ln in sys.stdin:
    m = re.search(r'remark: +(.*?) \[-Rpass', ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
import subprocess
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if not re.search(r'''$filt''', name): continue
    print(f\"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?')}\")
"
rm -f /tmp/kr_$$.o
