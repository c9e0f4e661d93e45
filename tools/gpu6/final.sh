#!/bin/bash
# round 6: state of the tree -- GPU suite, smoke, the default bench line (with its `configs` legs), rocprofv3 kernel stats of the
# same command (headline only), PMC passes -> profiles/pmc_traffic.json.   usage: bash tools/gpu6/final.sh <tag>
tag=${1:-r06_z}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -3 gpurun_out/${tag}_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; tail -1 gpurun_out/${tag}_smoke.txt
bash tools/profile_bench.sh $tag > /dev/null 2>&1
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
for l in open("gpurun_out/%s_bench.json" % tag):
    if l.startswith("{"):
        d = json.loads(l)
        print("headline ms/step %.3f tok/s %.1f" % (d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["kernels"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), round(d["roofline_step"]["frac"], 4))
        for c in d.get("configs", []):
            print("  ", c.get("label"), "| ms/step %.3f" % c.get("ms_per_step", 0), "frac %.3f" % c.get("frac", 0), {k: round(v, 1) for k, v in c.items() if k.endswith("_us")}, c.get("error"))
        print("  full_model", {k: v for k, v in d.get("full_model", {}).items() if k in ("tokens_per_s", "ms_per_token", "error")})
        print("  cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind")})
PY
cat gpurun_out/${tag}_kernel_stats.csv
PMC_OUT=/tmp timeout 600 bash tools/pmc_run.sh $tag python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > gpurun_out/${tag}_pmc_bench.txt 2>&1
python tools/pmc_traffic.py gpurun_out/${tag}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json
PMC_OUT=/tmp timeout 600 bash tools/pmc_run.sh ${tag}_cfg3 python bench.py --bits 3 --sinks 5 --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > gpurun_out/${tag}_cfg3_pmc_bench.txt 2>&1
python tools/pmc_traffic.py gpurun_out/${tag}_cfg3_pmc_bench.txt 3 131072 gpurun_out/pmc_traffic.json
