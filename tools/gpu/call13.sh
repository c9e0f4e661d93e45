#!/bin/bash
# round 3, GPU call 13: ablations of the p.V kernel (KVQ_V_DBG: 1 no look-up loop, 2 no row DMA, 4 DMA from an L2-resident
# source, 32 no chunk barrier), with and without the outlier phase
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c13
export TMPDIR=/tmp
for rep in 1 2; do
for sp in "" 1; do
for v in "" vdbg1 vdbg2 vdbg4 vdbg32 vdbg6; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  echo -n "nosparse=$sp "; KB_NOSPARSE=$sp KB_ONLY=mix_v KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done; done > ${O}_mixv.txt 2>&1
cat ${O}_mixv.txt
