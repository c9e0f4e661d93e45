#!/bin/bash
# round 4, call 16: PPL harness with the prompt-aware comparator (2048 tokens, 600 training steps, 3 draws), the GPU suite on
# the pruned kernels, default bench + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/ppl_delta.py 2048 600 3 > gpurun_out/r4c16_ppl.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4c16_tests.txt 2>&1
tail -14 gpurun_out/r4c16_ppl.txt | cut -c1-420; tail -5 gpurun_out/r4c16_tests.txt
