#!/usr/bin/env python3
"""Top kernels of a rocprofv3 kernel_stats.csv: calls, total ms, average us. usage: stats_top.py <kernel_stats.csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
for r in rows[:n]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['AverageNs']) / 1e3:9.1f} us")
