#!/bin/bash
# round 3, GPU call 7: compact formats, shard kernels, multi-process sharding tests, full suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c7
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > ${O}_tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline --compact > ${O}_bench_compact.json 2> ${O}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench.json 2>> ${O}_bench.err
cat ${O}_tests.txt
python - <<'PY'
import json
for f in ("gpurun_out/c7_bench_compact.json", "gpurun_out/c7_bench.json"):
    try:
        d = json.load(open(f)); print(f, "tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items()}, d["roofline"]["frac"])
    except Exception as e:
        print(f, e)
PY
tail -3 ${O}_bench.err
