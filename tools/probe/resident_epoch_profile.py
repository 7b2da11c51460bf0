#!/usr/bin/env python3
"""Where does the host spend a resident epoch of the product loop? cProfile over one steady-state epoch: time inside the final
collector's .cpu() = the host WAITING for the device (device-bound); everything else = host work per step. usage: [steps]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resident_epoch.py")).read().split("out = []")[0]
exec(src)
for _ in range(3):
    hh._train_each_epoch(loader, "train")
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
hh._train_each_epoch(loader, "train")
pr.disable()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(22)
txt = s.getvalue()
print("wall ms/step", round(1e3 * wall / nsteps, 3), "stats", hh.step_graph_stats)
print("\n".join(l[:150] for l in txt.splitlines()[4:34]))
