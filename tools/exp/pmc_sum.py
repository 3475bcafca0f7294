#!/usr/bin/env python3
"""Perf lab: per-dispatch counter values of the kernel with the largest counter totals in rocprofv3 --pmc csv trees.
usage: pmc_sum.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_lane_rows"
tot = defaultdict(float)
calls = defaultdict(set)
for path in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(path)):
        if want not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        calls[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(tot):
    n = len(calls[k])
    print(f"{k:28s} {tot[k] / n:16.0f}  per launch ({n} launches)")
