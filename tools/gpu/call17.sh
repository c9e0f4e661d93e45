#!/bin/bash
# round 3, GPU call 17: A/B on one box: p.V with outlier slabs (round-2 scheme, libkvq_oldv.so) vs without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c17
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in "" oldv; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=mix_v KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 32768 4096 2>&1 | grep -v "amdgpu.ids"
  KB_ONLY=mix_v KB_ITERS=150 timeout 300 python tools/kbench2.py 3 131149 32768 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_ab.txt 2>&1
unset KVQ_LIB
cat ${O}_ab.txt
