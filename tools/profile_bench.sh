#!/bin/bash
# usage (on the GPU box): tools/profile_bench.sh <tag> [bench.py args...]
# writes gpurun_out/<tag>_bench.json (unprofiled), gpurun_out/<tag>_bench_profiled.json and
# gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command, kvq kernels only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > gpurun_out/${tag}_bench_profiled.json 2> /tmp/prof_$tag.log
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" gpurun_out/${tag}_kernel_stats.csv <<'PY'
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
cat gpurun_out/${tag}_kernel_stats.csv
