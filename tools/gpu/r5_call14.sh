#!/bin/bash
# round 5, call 14: prefill pack with 4 histogram copies / word stores of the V codes: parity (bit-exact vs reference) + config 4 lines + prefill attention
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c14
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_atsize_gpu.py tests/test_fused_gpu.py tests/test_cache_gpu.py tests/test_prefill_attn_gpu.py tests/test_ties_gpu.py tests/test_head_shard_gpu.py -m gpu -q -x -k "prefill or pack or cache or ties or fused or shard" 2>&1 | tail -4 ) > ${O}_tests.txt
( timeout 600 python bench.py --prefill; timeout 600 python bench.py --prefill --bits 3 ) > ${O}_prefill.jsonl 2> ${O}_err.txt
timeout 300 python tools/prefill_attn_bench.py 8192 32768 > ${O}_attn.jsonl 2>> ${O}_err.txt
cat ${O}_tests.txt; python - <<'PY'
import json
for l in open("gpurun_out/r5c14_prefill.jsonl"):
    d = json.loads(l); print(d["config"]["label"], {k: round(v, 3) for k, v in d["kernels"].items()}, "frac", round(d["roofline"]["frac"], 3))
PY
cat ${O}_attn.jsonl; tail -3 ${O}_err.txt
