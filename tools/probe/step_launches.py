#!/usr/bin/env python3
"""Device launches of ONE eager optimizer step, by kernel: count and device time (torch.profiler, CUDA activity).
usage: step_launches.py [mode=abmil] [patches=8192] [bags=16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "abmil"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bags = int(sys.argv[3]) if len(sys.argv) > 3 else 16
case = bench.Case(torch, dev, kind, n, bags, max(16, bags), "bf16x3", seed=1, eager=True)
for _ in range(3):
    case.eager_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    case.eager_step()
    torch.cuda.synchronize()
agg = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        a = agg.setdefault(e.name.split("(")[0][:100], [0, 0.0]); a[0] += 1; a[1] += e.device_time
print("launches per step:", sum(v[0] for v in agg.values()), " device us:", round(sum(v[1] for v in agg.values()), 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[0]:3d}x {v[1]:8.1f} us  {k}")
