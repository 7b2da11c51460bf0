#!/usr/bin/env python3
"""Host profile of the PRODUCT loop's second epoch (every bag out of the device-resident cache): cProfile over
MyHandler._train_each_epoch on ragged pinned host bags, same construction as bench.py::product_loop.
usage: epoch_host_profile.py [steps]"""
import cProfile
import os
import pstats
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
if os.environ.get("PROBE_ODD"):            # real cohorts: the rows of a step batch are a multiple of 16, not of 256
    lens = [n + (16 * int(os.environ["PROBE_ODD"]) if i % 16 == 0 else 0) for i, n in enumerate(lens)]      # (negative: rows short of whole tiles)
print("rows of the first step batch:", sum(lens[:16]), "mod 256 =", sum(lens[:16]) % 256)
distinct = 64
pool = [torch.randn(1, lens[i], 1024, generator=g).pin_memory() for i in range(distinct)]
loader = [(torch.tensor([[i % distinct]], dtype=torch.int), [pool[i % distinct], torch.zeros(1, 1)],
           torch.tensor([[0.3 + 0.01 * (i % 50), float(i % 2)]])) for i in range(nbag)]
hh._train_each_epoch(loader, "train")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    hh._train_each_epoch(loader, "train")
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"resident epoch: host {1e3 * ti / nsteps:.3f} ms per step, wall {1e3 * ta / nsteps:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
hh._train_each_epoch(loader, "train")
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(45)

# un-profiled wall time of the loop's pieces (a blocking call shows up here, not under cProfile's CPU-ish view)
from advmil_amd import ingest, ops  # noqa: E402
acc = {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


for nm in ("begin", "add_device", "ready", "release", "batch_planes"):
    wrap(ingest.SlabStager, nm, "stager." + nm)
wrap(ingest.BagCache, "get", "cache.get")
for nm in ("_plan", "_update_disc", "_update_gen", "_disc_backward", "_disc_apply", "_gen_forward", "_gen_finish", "_slab", "_get_label_visiable_mask"):
    wrap(MyHandler, nm)
wrap(ops.Segments, "__init__", "Segments()")
t0 = time.perf_counter()
hh._train_each_epoch(loader, "train")
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print("epoch host %.3f ms per step; pieces (ms per step, nested ones overlap):" % (1e3 * tot / nsteps))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f" % (k, 1e3 * v / nsteps))
