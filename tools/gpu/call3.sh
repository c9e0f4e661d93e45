#!/bin/bash
# round 3, GPU call 3: wave priorities (s_setprio) in both matvecs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c3
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" r2base q1 q2 q3 q1p; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=score_k,mix_v timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_kbench.txt 2>&1
unset KVQ_LIB
for v in trk_q1 trk_q3 trk_q2; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_k.py 2>&1 | grep -v amdgpu.ids | tail -34; done > ${O}_trace_k.txt
for v in trv0 trv_q1 trv_q3 trk_q2; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_v.py 2>&1 | grep -v amdgpu.ids | tail -17; done > ${O}_trace_v.txt
cat ${O}_kbench.txt ${O}_trace_k.txt ${O}_trace_v.txt
