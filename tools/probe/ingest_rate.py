#!/usr/bin/env python3
"""Host side of the first-epoch ingest: a pageable loader tensor goes pageable -> pinned slab (CPU copy) -> HBM (DMA); a pinned one
goes straight to HBM. Rates per 8192 x 1024 fp32 bag (33.5 MB). usage: ingest_rate.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd.ingest import SlabStager  # noqa: E402

dev = torch.device("cuda", 0)
st = SlabStager(dev, 1024)
bags = [torch.randn(1, 8192, 1024) for _ in range(16)]
pinned = [b.pin_memory() for b in bags]
print("torch threads", torch.get_num_threads())
for name, src in (("pageable", bags), ("pinned", pinned)):
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.begin()
        for b in src:
            st.add(b)
        t1 = time.perf_counter()
        st.ready()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st.release()
    mb = 16 * 33.554432
    print(f"{name}: host {1e3 * (t1 - t0):.2f} ms for 16 bags ({mb / (t1 - t0) / 1e3:.1f} GB/s), until resident {1e3 * (t2 - t0):.2f} ms ({mb / (t2 - t0) / 1e3:.1f} GB/s)")
a = torch.empty(8192 * 1024)
b = torch.randn(8192 * 1024)
p = torch.empty(8192 * 1024).pin_memory()
for name, dst in (("pageable->pageable", a), ("pageable->pinned", p)):
    t0 = time.perf_counter()
    for _ in range(10):
        dst.copy_(b)
    dt = (time.perf_counter() - t0) / 10
    print(f"copy_ {name}: {1e3 * dt:.2f} ms per bag = {33.55 / dt / 1e3:.1f} GB/s")

# out-of-cache: 16 distinct source bags into one 537 MB destination
for name, dst in (("pageable slab", torch.empty(16 * 8192, 1024)), ("pinned slab", torch.empty(16 * 8192, 1024).pin_memory())):
    for rep in range(3):
        t0 = time.perf_counter()
        for i, bsrc in enumerate(bags):
            dst[i * 8192:(i + 1) * 8192].copy_(bsrc.reshape(-1, 1024))
        dt = time.perf_counter() - t0
    print(f"16 bags -> {name}: {1e3 * dt:.1f} ms ({16 * 33.55 / dt / 1e3:.1f} GB/s)")
import numpy as np
dstn = torch.empty(16 * 8192, 1024).pin_memory()
t0 = time.perf_counter()
for i, bsrc in enumerate(bags):
    np.copyto(dstn[i * 8192:(i + 1) * 8192].numpy(), bsrc.reshape(-1, 1024).numpy())
dt = time.perf_counter() - t0
print(f"numpy copyto -> pinned: {1e3 * dt:.1f} ms ({16 * 33.55 / dt / 1e3:.1f} GB/s)")
t0 = time.perf_counter()
for bsrc in bags:
    bsrc.is_pinned()
print(f"is_pinned() on a pageable 33.5 MB tensor: {1e3 * (time.perf_counter() - t0) / 16:.3f} ms per call")
t0 = time.perf_counter()
for bsrc in pinned:
    bsrc.is_pinned()
print(f"is_pinned() on a pinned tensor: {1e3 * (time.perf_counter() - t0) / 16:.3f} ms per call")

# the stager's pageable path, phase by phase
host = torch.empty(16 * 8192, 1024).pin_memory()
devb = torch.empty(16 * 8192, 1024, device=dev)
cs = torch.cuda.Stream()
for mode in ("copy then dma per bag", "all copies, then all dmas", "copies only"):
    for rep in range(3):
        torch.cuda.synchronize()
        tc = td = 0.0
        t0 = time.perf_counter()
        if mode == "copy then dma per bag":
            for i, bsrc in enumerate(bags):
                t = time.perf_counter()
                host[i * 8192:(i + 1) * 8192].copy_(bsrc.reshape(-1, 1024))
                tc += time.perf_counter() - t
                t = time.perf_counter()
                with torch.cuda.stream(cs):
                    devb[i * 8192:(i + 1) * 8192].copy_(host[i * 8192:(i + 1) * 8192], non_blocking=True)
                td += time.perf_counter() - t
        else:
            for i, bsrc in enumerate(bags):
                t = time.perf_counter()
                host[i * 8192:(i + 1) * 8192].copy_(bsrc.reshape(-1, 1024))
                tc += time.perf_counter() - t
            if mode != "copies only":
                t = time.perf_counter()
                with torch.cuda.stream(cs):
                    devb.copy_(host, non_blocking=True)
                td += time.perf_counter() - t
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
    print(f"{mode}: cpu copies {1e3 * tc:.1f} ms, dma issue {1e3 * td:.1f} ms, host total {1e3 * th:.1f}, until resident {1e3 * ta:.1f} ms")
