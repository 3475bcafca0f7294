#!/usr/bin/env python3
"""Prints the lines of a tools/bench_ops.py log as a table, and the per-kernel averages of a rocprofv3 stats file."""
import csv
import glob
import json
import sys

log = sys.argv[1]
for l in open(log):
    if l.startswith('{'):
        d = json.loads(l)
        if 'size_op' in d:
            print(f"{d['case']:4} q={d['queries']:<3} {d['size_op']:5} {d['size_ms']:8.4f} ms {d.get('size_GBps', 0):8.1f} GB/s | "
                  f"{d.get('elem_op', ''):5} {d.get('elem_ms', float('nan')):8.4f} ms {d.get('elem_GBps', float('nan')):8.1f} GB/s")
        else:
            print(d)
if len(sys.argv) > 2:
    for f in glob.glob(sys.argv[2] + '/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            name = r['Name']
            name = name.replace('sdrb::(anonymous namespace)::', '').replace('void ', '')
            print(f"{name[:90]:90} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}")
