#!/bin/bash
# round 4, GPU call 2: the affine p.V kernel -- correctness (new test file + the decode tests) and A/B timing against the
# per-row kernel (KVQ_MIX_ROWS=1) in bench.py, kernel stats under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c2
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mix_affine_gpu.py tests/test_decode_kv_gpu.py -x -q 2>&1 | tail -25 ) > ${O}_tests.txt
cat ${O}_tests.txt
for cfg in "--ctx 131072" "--ctx 32768" "--ctx 131072 --bits 3 --sinks 5"; do for v in 0 1; do
  KVQ_MIX_ROWS=$v timeout 600 python bench.py $cfg --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg rows=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['frac'])"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline > ${O}_prof.json 2> /tmp/prof_a.log
f=$(find /tmp/prof_a -name "*kernel_stats.csv" | head -1)
python - "$f" ${O}_kernel_stats.csv <<'PY'
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
cat ${O}_kernel_stats.csv
