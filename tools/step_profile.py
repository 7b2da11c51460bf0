#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of `bench.py --no-bf16-extra --no-roofline --no-cpu-baseline`:
the last N optimizer steps (delimited by the Adam launches) -> kernels/step, busy us/step, and the top kernels.
usage: step_profile.py out_kernel_trace.csv [steps=10] [top=30]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 10
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
a, b = adam[-2 * nst - 1] + 1, adam[-1] + 1
seq = rows[a:b]


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def short(n):
    n = re.sub(r"void at::native::|\(anonymous namespace\)::|at::native::", "", n)
    n = re.sub(r"vectorized_elementwise_kernel<4, ", "vec<", n)
    return n[:84]


wall = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3 / nst
busy = sum(dur(r) for r in seq) / nst
print(f"kernels/step {len(seq) / nst:.0f}  wall(profiled) {wall:.0f} us  busy {busy:.0f} us")
agg, cnt = collections.Counter(), collections.Counter()
for r in seq:
    agg[short(r["Kernel_Name"])] += dur(r)
    cnt[short(r["Kernel_Name"])] += 1
cum = 0.0
for n, v in agg.most_common(top):
    cum += v
    print(f"{n:84s} {cnt[n] / nst:6.1f}/step {v / nst:8.1f} us/step {v / cnt[n]:7.1f} us  cum {100 * cum / busy / nst:5.1f}%")
