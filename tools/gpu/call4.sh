#!/bin/bash
# round 3, GPU call 4: full GPU test-suite (new: ties vs the reference kernel, 128K end-to-end decode, prefill attention at
# 8K, PPL bars, CLI driver, one-call decode step), bench default + sweep with kvq_decode_step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c4
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > ${O}_tests.txt
timeout 900 python bench.py --no-cpu-baseline --no-fp16-baseline --sweep > ${O}_sweep.jsonl 2> ${O}_bench.err
KVQ_DECODE_MULTICALL=1 timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline --ctx 4096 --steps 20 > ${O}_4k_multicall.json 2>> ${O}_bench.err
cat ${O}_tests.txt; python - <<'PY'
import json
for f in ("gpurun_out/c4_sweep.jsonl", "gpurun_out/c4_4k_multicall.json"):
    for ln in open(f):
        try:
            d = json.loads(ln)
        except Exception:
            continue
        print(d["config"].get("label", "default"), "| tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: (round(v, 1) if v else v) for k, v in d["kernels"].items()})
PY
tail -5 ${O}_bench.err
