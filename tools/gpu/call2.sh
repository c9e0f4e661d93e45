#!/bin/bash
# round 3, GPU call 2: software-pipelined look-ups + staggered workgroup pairs in the score kernel; kernel timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c2
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" r2base p22 p22s14 p22s27 p22h14 p22h27 p22h40 j42s27 j42h27; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=score_k timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_kbench.txt 2>&1
unset KVQ_LIB
for v in trk_j42 trk_p22 trk_p22h27 trk_p22s27; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_k.py 2>&1 | grep -v amdgpu.ids | tail -40; done > ${O}_trace_k.txt
# correctness of the pipelined variant: the score / decode tests against it
cp kvquant_amd/libkvq.so /tmp/libkvq_keep.so; cp tools/abl/libkvq_p22h27.so kvquant_amd/libkvq.so
( timeout 900 python -m pytest tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > ${O}_tests_p22h27.txt
cp /tmp/libkvq_keep.so kvquant_amd/libkvq.so
( timeout 900 python -m pytest tests/test_decode_kv_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > ${O}_tests_default.txt
cat ${O}_kbench.txt ${O}_trace_k.txt ${O}_tests_p22h27.txt ${O}_tests_default.txt
