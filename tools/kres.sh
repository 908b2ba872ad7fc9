#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra hipcc flags]  -> one line per kernel: name, VGPRs, AGPRs, SGPRs, scratch, occupancy, LDS
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re
cur = {}
for ln in sys.stdin:
    m = re.search(r"remark: +([A-Za-z \[\]/]+): (.+?) \[-Rpass", ln)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if cur: print(cur)
        cur = {"fn": v[:70]}
    elif k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill", "SGPRs Spill"):
        cur[k.replace(" [bytes/lane]","").replace(" [waves/SIMD]","").replace(" [bytes/block]","")] = v
if cur: print(cur)
'
