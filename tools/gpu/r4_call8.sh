#!/bin/bash
# round 4, GPU call 8: fused decode kernel correctness after the -inf fix
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c8
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_fused_decode_gpu.py -q 2>&1 | tail -25 ) > ${O}_tests.txt
cat ${O}_tests.txt
