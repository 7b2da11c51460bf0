#!/usr/bin/env python3
"""Throughput of the supervised baseline step (advmil_amd.model.BaselineHandler._update_network): bags/s over resident bags,
eager launches (the step has ~10x fewer launches than the G+D step). usage: baseline_bench.py [--mode abmil] [--task surv_reg]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import synth  # noqa: E402
from advmil_amd.config import default_baseline_cfg  # noqa: E402
from advmil_amd.model import BaselineHandler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="abmil", choices=["abmil", "patch", "cluster"])
ap.add_argument("--task", default="surv_reg", choices=["surv_reg", "surv_nll", "surv_cox"])
ap.add_argument("--bags", type=int, default=16)
ap.add_argument("--patches", type=int, default=8192)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--gemm-mode", default="bf16x3")
a = ap.parse_args()
dev = torch.device("cuda", 0)
pdh = "384-4" if a.task == "surv_nll" else "384-1"
h = BaselineHandler(default_baseline_cfg(bcb_mode=a.mode, task=a.task, pdh_dims=pdh, bp_every_batch=a.bags, gemm_mode=a.gemm_mode), device=dev)
npool = 4 * a.bags
slab = torch.randn(npool * a.patches, 1024, device=dev)
xs = [[slab[i * a.patches:(i + 1) * a.patches].view(1, a.patches, 1024),
       torch.from_numpy(synth.cluster_ids(0, i, a.patches)).to(dev) if a.mode == "cluster" else torch.zeros(1, 1, device=dev)] for i in range(npool)]
ys = []
for i in range(npool):
    t = (i % 4) if a.task == "surv_nll" else (0.1 + 0.8 * ((i * 37) % 100) / 100.0) * (100.0 if a.task == "surv_cox" else 1.0) + i * 1e-3
    ys.append(torch.tensor([[float(t), float(i % 2)]], device=dev))


def step(k):
    g0 = (k % 4) * a.bags
    h._update_network(k, xs[g0:g0 + a.bags], ys[g0:g0 + a.bags])


for k in range(3):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(a.steps):
    step(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
h.pop_logs()
print(f"baseline {a.mode}/{a.task} {a.gemm_mode}: {a.bags * a.steps / dt:.1f} bags/s, {1e3 * dt / a.steps:.2f} ms per {a.bags}-bag step (eager)")
