#!/usr/bin/env python3
"""Ordered launch list of ONE optimizer step from a rocprofv3 --kernel-trace CSV (the last step, delimited by the Adam launches):
start offset, duration and the idle gap in front of every launch. usage: step_timeline.py out_kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
a, b = adam[-3] + 1, adam[-1] + 1
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0.0
for i, r in enumerate(seq):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"void at::native::|\(anonymous namespace\)::|at::native::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:70]
    g = (s - prev_end) / 1e3
    gaps += max(g, 0.0)
    busy += (e - s) / 1e3
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {g:6.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8} wg {r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')):>5}  {n}")
    prev_end = max(prev_end, e)
print(f"launches {len(seq)}  wall {(prev_end - t0) / 1e3:.1f} us  busy {busy:.1f} us  gaps {gaps:.1f} us")
