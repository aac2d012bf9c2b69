#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 PMC counter from a `--pmc X --output-format csv` run.
Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> > profiles/x_pmc_X.txt"""
import csv, glob, os, sys
from collections import defaultdict

files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: [0, 0.0])
name = None
for f in files:
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"], row["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
for (kern, ctr), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-72s launches %6d  %s avg %.1f (raw counter units per launch)" % (kern[:72], n, ctr, tot / n))
