#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for L in 4096 32768; do KVQ_LIB=tools/abl/libkvq_trk.so timeout 200 python tools/dbg/trace_k_short.py $L 2>&1 | grep -v amdgpu; done > gpurun_out/r4c27_trace.txt
cat gpurun_out/r4c27_trace.txt
