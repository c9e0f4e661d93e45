#!/bin/bash
# round 3, GPU call 9: which unit bounds the score kernel?  ablation builds (tools/abl/build_abl.sh; KVQ_ABL bits:
# 1 no LDS look-ups, 2 no per-head barrier, 4 no table DMA after the first two, 64 no dense section, 128 no outlier step,
# 256 no packed-word re-loads, 512 no outlier-entry loads) + the load-width microbenchmark
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c9
export TMPDIR=/tmp
timeout 120 tools/ubench/load_rate > ${O}_load_rate.txt 2>&1
for rep in 1 2; do
for v in "" 64 192 708 256 772 1 128 2 900 902; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_abl$v.so; fi
  KB_ONLY=score_k KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_abl.txt 2>&1
cat ${O}_load_rate.txt ${O}_abl.txt
