#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a
short CSV: name (truncated), calls, total_us, avg_us, pct.
usage: python tools/prof_summary.py <results.db> <out.csv> [filter-substring]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)            # drop argument list
    name = name.replace("void ", "")
    return name[:110]


def main():
    db, out = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            if flt and flt not in name:
                continue
            w.writerow([short(name), calls, "%.1f" % tot, "%.2f" % avg, "%.2f" % pct])
    print("wrote", out)


if __name__ == "__main__":
    main()
