#!/bin/bash
# round 5, last call: the final tree once more -- default bench + rocprofv3 kernel stats of the same command, PMC passes -> HBM traffic
# per launch for the final kernel hash, config 4 (prefill) under rocprofv3, the full GPU test suite, smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=r05_y; O=gpurun_out/$T
export TMPDIR=/tmp
bash tools/profile_bench.sh $T > ${O}_profile.log 2>&1
PMC_OUT=/tmp bash tools/pmc_run.sh $T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_pmc_bench.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_traffic.py ${O}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json > ${O}_pmc_traffic.log 2>&1
# config 4 under rocprofv3: pack K / V / MFMA attention kernel durations
rm -rf /tmp/prof_pf; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -o p -- python $GRAFT_REPO_ROOT/bench.py --prefill --bits 4 > $GRAFT_REPO_ROOT/${O}_prefill_profiled.json 2> /tmp/prof_pf.log; cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_pf -name "*kernel_stats.csv" | head -1)
python - "$f" ${O}_prefill_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = csv.writer(open(sys.argv[2], "w", newline=""))
out.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if "kvq::" not in name:
        continue
    out.writerow([name[:110], r["Calls"], "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"]])
PY
python bench.py --prefill --bits 4 2>/dev/null > ${O}_prefill.json
python bench.py --prefill --bits 3 2>/dev/null >> ${O}_prefill.json
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > ${O}_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.txt 2>&1
head -c 1800 ${O}_bench.json; echo; cat ${O}_kernel_stats.csv; cat ${O}_prefill_kernel_stats.csv; tail -3 ${O}_gpu_tests.txt; tail -5 gpurun_out/pmc_traffic.json; tail -1 ${O}_smoke.txt
