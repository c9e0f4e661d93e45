#!/bin/bash
# round 4, call 15: PPL harness on the trained model, the reference-attention replay, ties, at-size sinks, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/ppl_delta.py 1024 300 > gpurun_out/r4c15_ppl.txt 2>&1
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_ties_gpu.py tests/test_atsize_gpu.py -m gpu -q -x > gpurun_out/r4c15_tests.txt 2>&1
timeout 600 python bench.py > gpurun_out/r4c15_bench.json 2> gpurun_out/r4c15_bench.err
tail -5 gpurun_out/r4c15_ppl.txt; tail -5 gpurun_out/r4c15_tests.txt; cat gpurun_out/r4c15_bench.json; tail -3 gpurun_out/r4c15_bench.err
