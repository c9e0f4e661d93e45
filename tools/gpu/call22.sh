#!/bin/bash
# round 3, GPU call 22: head-group planner of the score kernel, round-counting model (libkvq_oldplan.so) vs makespan model
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c22
export TMPDIR=/tmp
for v in "" oldplan; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=score_k KB_ITERS=100 timeout 600 python tools/kbench2.py 4 2048 4096 8192 16384 24576 32768 49152 65536 81920 98304 114688 131149 163840 196608 262144 2>&1 | grep -v "amdgpu.ids"
done > ${O}_plan.txt 2>&1
cat ${O}_plan.txt
