#!/bin/bash
# round 4, call 26: what the driver runs at round end -- the GPU suite, smoke(), the default bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r04_b
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > ${O}_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.txt 2>&1
timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err
tail -3 ${O}_gpu_tests.txt; tail -2 ${O}_smoke.txt; head -c 2500 ${O}_bench.json
