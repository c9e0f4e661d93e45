#!/bin/bash
# round 6, item 7: quant_cuda's _opt2 over the shadow mirror -- parity (test_ref_gpu, at-size), the reference's kernels next to
# libkvq in the decode pattern (tools/ref_bench.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ref_gpu.py tests/test_atsize_gpu.py -x -q -m gpu > gpurun_out/r06_s_tests.txt 2>&1; tail -5 gpurun_out/r06_s_tests.txt
for b in 4 3; do timeout 600 python tools/ref_bench.py $b 4096 32768 131072; done > gpurun_out/r06_s_ref_bench.jsonl 2> gpurun_out/r06_s_ref_bench.err
cat gpurun_out/r06_s_ref_bench.jsonl; tail -3 gpurun_out/r06_s_ref_bench.err
