#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4c32_tests.txt
cat gpurun_out/r4c32_tests.txt
