#!/bin/bash
# round 5, call 3: phase timeline of the p.V kernel with the new outlier phase
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c3
export TMPDIR=/tmp
( echo "== non-fused"; KVQ_LIB=tools/abl/libkvq_vtrace.so timeout 300 python tools/dbg/trace_v.py; echo "== fused"; FUSED=1 KVQ_LIB=tools/abl/libkvq_vtrace.so timeout 300 python tools/dbg/trace_v.py ) > ${O}_trace.txt 2>&1
cat ${O}_trace.txt
