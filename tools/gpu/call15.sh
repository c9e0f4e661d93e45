#!/bin/bash
# round 3, GPU call 16: p.V without outlier slabs -- tests, sweep, PMC traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c16
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_fuzz_gpu.py tests/test_compact_gpu.py tests/test_ties_gpu.py tests/test_fullsize_gpu.py tests/test_atsize_gpu.py tests/test_sharding_gpu.py -m gpu -q 2>&1 | tail -8 ) > ${O}_tests.txt
timeout 900 python bench.py --sweep --no-cpu-baseline --no-fp16-baseline > ${O}_sweep.jsonl 2> ${O}_sweep.err
PMC_OUT=/tmp timeout 900 bash tools/pmc_run.sh c16 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline > ${O}_pmc.txt 2>&1
cat ${O}_tests.txt
python - <<'PY'
import json
for l in open("gpurun_out/c16_sweep.jsonl"):
    d = json.loads(l); c = d["config"]
    print(c.get("label", ""), c["ctx"], c["bits"], c.get("outlier_format", "")[:8], "tok/s %.1f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})
PY
grep -A 24 "mix_v_kernel<4" ${O}_pmc.txt | grep -E "mix_v|FETCH|WRITE|INSTS_VALU|INSTS_LDS" | head; grep -A 24 "mix_v_reduce" ${O}_pmc.txt | grep -E "FETCH|WRITE" | head -3
