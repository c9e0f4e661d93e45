#!/bin/bash
# round 4, call 30 (after the tail batching of the score kernel): the profiles of the round -- default bench (+ full model leg) and rocprofv3 kernel stats of the same command,
# PMC passes -> HBM traffic per launch stamped with the kernels' code hash, the sweep, the GPU test log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=r04_c; O=gpurun_out/$T
export TMPDIR=/tmp
bash tools/profile_bench.sh $T > ${O}_profile.log 2>&1
PMC_OUT=/tmp bash tools/pmc_run.sh $T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_pmc_bench.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_traffic.py ${O}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json > ${O}_pmc_traffic.log 2>&1
timeout 1500 python bench.py --no-cpu-baseline --no-fp16-baseline --no-full-model --sweep 2> ${O}_sweep.err | head -n 9 > ${O}_sweep.jsonl
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > ${O}_gpu_tests.txt
head -c 3500 ${O}_bench.json; echo; cat ${O}_kernel_stats.csv; tail -4 ${O}_gpu_tests.txt; cat gpurun_out/pmc_traffic.json; python - <<'PY'
import json
for l in open("gpurun_out/r04_c_sweep.jsonl"):
    d = json.loads(l); print(d["config"].get("label"), "%.3f ms/step %.1f tok/s" % (d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")}, round(d["roofline"]["frac"], 3))
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.txt 2>&1; tail -1 ${O}_smoke.txt
