#!/bin/bash
# round 5, call 12: full GPU suite + default bench + prefill config line on the current tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c12
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > ${O}_gpu_tests.txt
timeout 600 python bench.py --no-full-model > ${O}_bench.json 2> ${O}_bench.err
timeout 600 python bench.py --prefill > ${O}_prefill4.json 2>> ${O}_bench.err
timeout 600 python bench.py --prefill --bits 3 > ${O}_prefill3.json 2>> ${O}_bench.err
tail -5 ${O}_gpu_tests.txt; python - <<'PY'
import json
for f in ("r5c12_bench.json", "r5c12_prefill4.json", "r5c12_prefill3.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, "%.3f ms/step %.1f %s" % (d["ms_per_step"], d["value"], d["unit"]), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["kernels"].items()}, "frac", round(d["roofline"]["frac"], 3), "step frac", round(d.get("roofline_step", {}).get("frac", 0), 3))
        if "cpu_baseline" in d: print("   cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 ${O}_bench.err
