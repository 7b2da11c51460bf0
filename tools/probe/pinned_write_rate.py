#!/usr/bin/env python3
"""CPU write rate into a pinned (hipHostMalloc) buffer before and after the device has read it. usage: pinned_write_rate.py"""
import time

import torch

dev = torch.device("cuda", 0)
torch.cuda.init()
bags = [torch.randn(8192, 1024) for _ in range(16)]


def fill(dst, tag):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i, b in enumerate(bags):
            dst[i * 8192:(i + 1) * 8192].copy_(b)
        best = min(best, time.perf_counter() - t0)
    print(f"{tag}: {1e3 * best:.1f} ms for 16 bags ({16 * 33.55 / best / 1e3:.1f} GB/s)", flush=True)


pg = torch.empty(16 * 8192, 1024)
fill(pg, "pageable destination")
pin = torch.empty(16 * 8192, 1024).pin_memory()
fill(pin, "pinned, never read by the device")
devb = torch.empty(16 * 8192, 1024, device=dev)
devb.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
fill(pin, "pinned, after one H2D from it")
fill(pin, "pinned, again")
pin2 = torch.empty(16 * 8192, 1024, pin_memory=True)
fill(pin2, "second pinned buffer, never read by the device")
fill(pg, "pageable destination again")
torch.set_num_threads(16)
fill(pin, "pinned (after H2D), 16 threads")
fill(pin2, "second pinned, 16 threads")
torch.set_num_threads(128)


def fill2(dst, tag, sync):
    for rep in range(3):
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, b in enumerate(bags):
            dst[i * 8192:(i + 1) * 8192].copy_(b)
        dt = time.perf_counter() - t0
    print(f"{tag}: {1e3 * dt:.1f} ms (last of 3)", flush=True)


fill2(pin, "pinned, no sync before", False)
fill2(pin, "pinned, torch.cuda.synchronize() before each rep", True)
bags = [b.reshape(1, 8192, 1024) for b in bags]


def fill3(dst, tag):
    for rep in range(3):
        t0 = time.perf_counter()
        for i, b in enumerate(bags):
            dst[i * 8192:(i + 1) * 8192].copy_(b.reshape(-1, 1024))
        dt = time.perf_counter() - t0
    print(f"{tag}: {1e3 * dt:.1f} ms (last of 3)", flush=True)


fill3(pin, "pinned, 3-D sources reshaped")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd.ingest import SlabStager
st = SlabStager(dev, 1024)
for rep in range(3):
    t0 = time.perf_counter()
    st.begin()
    for b in bags:
        st.add(b)
    t1 = time.perf_counter()
    st.ready(); torch.cuda.synchronize(); st.release()
    print(f"stager.add x16 pageable: {1e3 * (t1 - t0):.1f} ms", flush=True)
fill3(pin, "pinned again after the stager ran")
