#!/usr/bin/env python3
"""Evaluation pass (MyHandler.test_model out of the bag cache) against the number of bags per slab. usage: eval_batch_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd.config import default_cfg  # noqa: E402
from advmil_amd.model import MyHandler  # noqa: E402

dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "abmil"
base = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
hh = MyHandler(default_cfg(bcb_mode=mode, bp_every_batch=16, cuda_id=0, gemm_mode="bf16x3"), device=dev)
g = torch.Generator().manual_seed(7)
fr = (0.75, 1.0, 1.25, 0.5, 1.5, 1.0, 0.875, 1.125)
n = 128
items = [(torch.tensor([[i]], dtype=torch.int), [torch.randn(1, int(base * fr[i % 8]) // 16 * 16, 1024, generator=g).pin_memory(), torch.zeros(1, 1)],
          torch.tensor([[0.3, 1.0]])) for i in range(n)]


class DS:
    pass


class DL:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        return iter(items)


dl = DL(DS())
for nb in (8, 16, 32, 64):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = MyHandler.test_model(hh.netG, hh.netD, mode, dl, times_test_sample=1, batch_bags=nb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode} {base}: {nb} bags per slab: {n / dt:.0f} bags/s ({1e3 * dt / n:.3f} ms per bag)", flush=True)
