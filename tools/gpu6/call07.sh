#!/bin/bash
# round 6, call 7: state of the tree: GPU suite, default bench line (with the `configs` legs), rocprofv3 kernel stats, PMC traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_g_gpu_tests.txt 2>&1
tail -4 gpurun_out/r06_g_gpu_tests.txt
bash tools/profile_bench.sh r06_g > /dev/null 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_g_bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("headline ms/step %.3f tok/s %.1f" % (d["ms_per_step"], d["value"]), d["kernels"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline_step"]["frac"])
        for c in d.get("configs", []):
            print("  ", c.get("label"), "| ms/step", c.get("ms_per_step"), "frac", c.get("frac"), {k: round(v, 1) for k, v in c.items() if k.endswith("_us")}, c.get("error"))
        print("  full_model", d.get("full_model"))
PY
cat gpurun_out/r06_g_kernel_stats.csv
PMC_OUT=/tmp bash tools/pmc_run.sh r06_g python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > gpurun_out/r06_g_pmc_bench.txt 2>&1
grep -A22 "mix_v_wide_kernel<4\|score_k_kernel<4" gpurun_out/r06_g_pmc_bench.txt | head -80
