#!/bin/bash
# round 3, GPU call 24: fused softmax as the default at every length: tests + sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c24
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > ${O}_tests.txt
timeout 1200 python bench.py --sweep --no-cpu-baseline --no-fp16-baseline > ${O}_sweep.jsonl 2> ${O}_sweep.err
cat ${O}_tests.txt
python - <<'PY'
import json
for l in open("gpurun_out/c24_sweep.jsonl"):
    d = json.loads(l); c = d["config"]
    print(c.get("label", ""), c["ctx"], c["bits"], c.get("outlier_format", "")[:8], "tok/s %.1f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY
