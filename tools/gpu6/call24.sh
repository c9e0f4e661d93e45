#!/bin/bash
# round 6: the select paths against the torch specification (tests/test_select_gpu.py), then the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_select_gpu.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_x_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_x_gpu_tests.txt
