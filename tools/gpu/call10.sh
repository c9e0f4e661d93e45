#!/bin/bash
# round 3, GPU call 10: the dense section alone (KVQ_ABL=902: no loads, no outlier step, no barrier) under different
# look-up schedules: does any of them lower the floor that the ablation of call 9 found (68-70 us)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c10
export TMPDIR=/tmp
for rep in 1 2; do
for v in abl902 d_p22 d_44 d_22 d_84 d_p42; do
  export KVQ_LIB=tools/abl/libkvq_$v.so
  KB_ONLY=score_k KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_dense.txt 2>&1
cat ${O}_dense.txt
