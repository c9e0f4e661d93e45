#!/bin/bash
# round 3, GPU call 28: what the driver runs at round end -- smoke(), the GPU test suite, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c28
export TMPDIR=/tmp
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > ${O}_smoke.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > ${O}_tests.txt
timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err
cat ${O}_smoke.txt ${O}_tests.txt; tail -3 ${O}_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/c28_bench.json")); print("tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
