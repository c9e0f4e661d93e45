#!/bin/bash
# round 6, call 1: same-box baseline of the round-5 tree (default line, config sweep) + the atomic-add rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
tools/ubench/atomic_rate > gpurun_out/r06_atomic_rate.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model > gpurun_out/r06_a_base_bench.json 2> gpurun_out/r06_a_base_bench.err
python bench.py --sweep --no-cpu-baseline --no-fp16-baseline --no-full-model > gpurun_out/r06_a_base_sweep.jsonl 2> gpurun_out/r06_a_base_sweep.err
cat gpurun_out/r06_atomic_rate.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06_a_base_bench.json", "gpurun_out/r06_a_base_sweep.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d.get("config", {}).get("workload"), "| ms/step %.3f" % d["ms_per_step"], d.get("kernels"), (d.get("roofline") or {}).get("frac"))
PY
