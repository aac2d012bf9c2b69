#!/usr/bin/env python3
"""Compact view of a rocprofv3 `*kernel_stats.csv`: python tools/kstats.py file.csv [min_percent] -- kernel names cut to 70 characters
(torch's templated names run to kilobytes), one line per kernel."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
print("%-72s %7s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows:
    if float(r["Percentage"]) < thr:
        continue
    name = re.sub(r"^void ", "", r["Name"])
    name = re.sub(r"\(.*$", "", name)[:70]
    print("%-72s %7d %12.1f %10.2f %7.2f" % (name, int(r["Calls"]), int(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
