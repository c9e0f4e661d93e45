#!/usr/bin/env python
"""profiles/pmc_traffic.json from a tools/pmc_run.sh summary: HBM bytes per launch of the two matvecs =
2 x FETCH_SIZE + WRITE_SIZE (KiB; the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).
usage: python tools/pmc_traffic.py <summary.txt> <bits> <ctx> [out.json]"""
import json
import re
import sys

src, bits, ctx = sys.argv[1], sys.argv[2], sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
cur, vals = None, {}
for line in open(src):
    if line.startswith("kvq::"):
        cur = line.strip()
        vals[cur] = {}
    elif cur:
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m:
            vals[cur][m.group(1)] = float(m.group(2))
try:
    j = json.load(open(out))
except Exception:
    j = {}
for key, pat in (("score_k", r"score_k_kernel"), ("mix_v", r"mix_v(_wide)?_kernel")):
    ks = [k for k in vals if re.search(pat, k) and "FETCH_SIZE" in vals[k]]
    if not ks:
        continue
    k = max(ks, key=lambda n: vals[n]["FETCH_SIZE"])
    b = int((2 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024)
    if key == "mix_v":
        r = [n for n in vals if "mix_v_reduce" in n]
        if r:
            b += int((2 * vals[r[0]]["FETCH_SIZE"] + vals[r[0]]["WRITE_SIZE"]) * 1024)
    j.setdefault(key, {})["%s_%s" % (bits, ctx)] = b
    if bits == "4":
        j[key][str(ctx)] = b
j["_source"] = src
sys.path.insert(0, ".")
from kvquant_amd import build as kb  # noqa: E402
j["_kernel_code_sha"] = kb.kernel_source_hash()      # bench.py drops the numbers when the kernels' code has changed since
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j))
