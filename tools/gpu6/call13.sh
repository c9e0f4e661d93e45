#!/bin/bash
# round 6, call 13: the wide p.V kernel forced at every length -- oracle suites, head shards, odd shapes (tests/test_wide_gpu.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KVQ_V_WIDE_FROM=1 timeout 900 python -m pytest tests/wide_shapes_check.py -q -m gpu > gpurun_out/r06_m_wide_shapes.txt 2>&1
tail -5 gpurun_out/r06_m_wide_shapes.txt
timeout 1500 python -m pytest tests/test_wide_gpu.py -x -q -m gpu > gpurun_out/r06_m_wide_tests.txt 2>&1
tail -5 gpurun_out/r06_m_wide_tests.txt
