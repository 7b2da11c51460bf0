#!/usr/bin/env python3
"""Where does the HOST time of an eager optimizer step go? cProfile over eager steps on resident bags (no PCIe), top functions by
own time and by cumulative time; plus wall per step with the GPU kept busy vs the host-only issue time. usage: eager_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
case = bench.Case(torch, dev, "abmil", 8192, 16, 32, "bf16x3", 1, eager=True)
for _ in range(5):
    case.eager_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    case.eager_step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"eager: host issue {1e3 * t_issue / steps:.3f} ms per step, wall {1e3 * t_all / steps:.3f} ms per step")
# pure host cost of the same launch sequence: tiny bags, so that the device is never the limit
small = bench.Case(torch, dev, "abmil", 4096, 16, 16, "bf16x3", 1, eager=True) if os.environ.get("PROBE_SMALL", "1") == "1" else None
if small is not None:
    for mt in (True, False):
        torch.autograd.set_multithreading_enabled(mt)
        for _ in range(5):
            small.eager_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            small.eager_step()
        ti = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
        print(f"4096-patch bags, autograd multithreading {mt}: host issue {1e3 * ti / steps:.3f} ms per step, wall {1e3 * ta / steps:.3f}")
    small.free()
torch.autograd.set_multithreading_enabled(os.environ.get("PROBE_MT", "0") == "1")      # backward on THIS thread: cProfile sees it
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    case.eager_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
