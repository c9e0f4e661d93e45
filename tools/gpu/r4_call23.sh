#!/bin/bash
# round 4, call 23: where the full-model leg's time goes (rocprofv3 kernel stats of a run whose timed part is mostly that leg)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c23
rm -rf /tmp/prof_fm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fm -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline > ${O}_bench.json 2> ${O}_err.txt
f=$(find /tmp/prof_fm -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > ${O}_stats.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    print("%-100s calls %6s total_us %10.1f avg_us %8.2f" % (name[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
cat ${O}_stats.txt; python -c "
import json; d=json.load(open('${O}_bench.json')); print(d['full_model'])"
