#!/bin/bash
# round 3, GPU call 11: what do the packed-word loads cost, and why?  tm255: the same loads from an L2-resident source;
# a1024: half of them; a512: no outlier-entry loads; a4: no table DMA; a260: neither words nor tables
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c11
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in "" tm255 a1024 a1024tm a512 a4 a260 abl256; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=score_k KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_loads.txt 2>&1
cat ${O}_loads.txt
