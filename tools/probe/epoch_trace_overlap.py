#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv of tools/probe/epoch_host_profile.py and reports, for the resident epochs: how much of the
stage_bag_kernel time (copy stream) lies under compute kernels of other streams, and the device busy time per optimizer step.
usage: epoch_trace_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in rows]
ks.sort()
stage = [k for k in ks if "stage_bag" in k[2]]
comp = [k for k in ks if "stage_bag" not in k[2]]
print("kernels", len(ks), "stage_bag", len(stage), "queues", sorted({k[3] for k in ks}), "stage queues", sorted({k[3] for k in stage}))
if not stage:
    sys.exit()
# union of compute intervals
iv = []
for s, e, *_ in comp:
    if iv and s <= iv[-1][1]:
        iv[-1][1] = max(iv[-1][1], e)
    else:
        iv.append([s, e])
import bisect
starts = [a for a, _ in iv]
tot = und = 0
for s, e, *_ in stage:
    tot += e - s
    i = max(bisect.bisect_right(starts, s) - 1, 0)
    while i < len(iv) and iv[i][0] < e:
        und += max(0, min(e, iv[i][1]) - max(s, iv[i][0]))
        i += 1
print("stage_bag total %.1f ms, avg %.1f us; under compute kernels %.1f %%" % (tot / 1e6, tot / len(stage) / 1e3, 100.0 * und / tot))
t0, t1 = stage[len(stage) // 3][0], stage[-1][1]
busy = sum(min(b, t1) - max(a, t0) for a, b in iv if b > t0 and a < t1)
nst = sum(1 for k in stage if t0 <= k[0] <= t1) / 16.0
print("window %.1f ms, ~%.1f steps, compute-union busy %.3f ms per step, window %.3f ms per step" % ((t1 - t0) / 1e6, nst, busy / 1e6 / nst, (t1 - t0) / 1e6 / nst))
