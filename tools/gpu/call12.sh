#!/bin/bash
# round 3, GPU call 12: shader clock and phase times of the score kernel with and without its HBM traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c12
export TMPDIR=/tmp
for v in trk trk_tm trk_256 trk_902; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_k.py 2>&1 | grep -v amdgpu.ids | head -22; done > ${O}_clock.txt
cat ${O}_clock.txt
