#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + PMC passes) per kernel name."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void cwt::", "").replace("cwt::", "")[:60]


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print(f"{short(row['Name']):60s} calls {row['Calls']:>6s} total_ns {row['TotalDurationNs']:>12s} "
              f"avg_ns {float(row['AverageNs']):12.1f} pct {row['Percentage']}")

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("== no counter csv in", d)
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    print("== per-dispatch averages from", os.path.basename(d))
    for k in agg:
        vals = "  ".join(f"{c}={agg[k][c] / cnt[k][c]:.4g}" for c in sorted(agg[k]))
        n = max(cnt[k].values())
        print(f"{k:60s} n={n:5d}  {vals}")
