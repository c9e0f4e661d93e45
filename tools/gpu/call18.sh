#!/bin/bash
# round 3, GPU call 18: whole GPU suite + default bench after the p.V change and the driver additions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c18
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > ${O}_tests.txt
timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err
cat ${O}_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/c18_bench.json")); print("tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
