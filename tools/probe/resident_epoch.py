#!/usr/bin/env python3
"""Wall time per step of the PRODUCT loop's resident epochs (MyHandler._train_each_epoch, every bag out of the device cache), same
construction as bench.py::product_loop; for A/B runs of the staging path under environment switches. usage: resident_epoch.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd.config import default_cfg  # noqa: E402
from advmil_amd.model import MyHandler  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bags, base = 16, 8192
dev = torch.device("cuda", 0)
hh = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=bags, cuda_id=0, gemm_mode="bf16x3"), device=dev)
g = torch.Generator().manual_seed(7)
fr = (0.75, 1.0, 1.25, 0.5, 1.5, 1.0, 0.875, 1.125)
nbag = bags * nsteps
lens = [int(base * fr[i % len(fr)]) // 16 * 16 for i in range(nbag)]
distinct = 64
pool = [torch.randn(1, lens[i], 1024, generator=g).pin_memory() for i in range(distinct)]
loader = [(torch.tensor([[i % distinct]], dtype=torch.int), [pool[i % distinct], torch.zeros(1, 1)],
           torch.tensor([[0.3 + 0.01 * (i % 50), float(i % 2)]])) for i in range(nbag)]


class _DS:
    def __init__(self, items):
        self.items = items


class _DL:                                   # (a loader with a `.dataset`: the scope of the device-resident bag cache)
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        return iter(self.dataset.items)


loader = _DL(_DS(loader))
hh._train_each_epoch(loader, "train")
torch.cuda.synchronize()
out = []
for rep in range(5):
    t0 = time.perf_counter()
    hh._train_each_epoch(loader, "train")
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    out.append((1e3 * ti / nsteps, 1e3 * ta / nsteps))
print("resident epoch (host, wall) ms per step:", "  ".join(f"{a:.3f}/{b:.3f}" for a, b in out), " env:",
      {k: v for k, v in os.environ.items() if k.startswith("ADVMIL_") and k != "ADVMIL_HIP_LIB"}, flush=True)
