#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra flags]   -> one line per kernel: name vgpr sgpr occupancy lds scratch
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -fvisibility=hidden -DKVQ_BUILD "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | python3 -c '
import sys,re
cur=None
def out(c):
    if c: print("%-75s vgpr %-4s sgpr %-4s occ %-2s lds %-7s scratch %s spill %s" % (c["name"][:75],c.get("VGPRs"),c.get("TotalSGPRs"),c.get("Occupancy"),c.get("LDS"),c.get("ScratchSize"),c.get("VGPRsSpill")))
for l in sys.stdin:
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass",l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        out(cur); cur={"name":v}
    elif cur is not None:
        cur[k.replace(" ","").replace("Size","Size") if k!="LDS Size" else "LDS"]=v
        if k=="LDS Size": cur["LDS"]=v
        if k=="VGPRs Spill": cur["VGPRsSpill"]=v
out(cur)
'
