#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel over the passes written by tools/pmc_run.sh."""
import csv
import glob
import re
import sys
from collections import defaultdict

prefix, n = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for i in range(n):
    for f in glob.glob("%s%d/**/*counter_collection.csv" % (prefix, i), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", "")).replace("void ", "")[:60]
            if "kvq::" not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
