#!/usr/bin/env python3
"""pageable -> pinned host copy of 16 x 33.5 MB bags: torch copy_ (OpenMP) vs numpy copyto from a small thread pool. usage: host_copy_threads.py"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

torch.cuda.init()
bags = [torch.randn(1, 8192, 1024) for _ in range(16)]
pin = torch.empty(16 * 8192, 1024).pin_memory()
pv = pin.numpy()


def rep(fn, tag):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f"{tag}: min {1e3 * min(ts):.1f} ms, max {1e3 * max(ts):.1f} ms ({16 * 33.55 / min(ts) / 1e3:.0f} GB/s best)", flush=True)


def torch_copy():
    for i, b in enumerate(bags):
        pin[i * 8192:(i + 1) * 8192].copy_(b.reshape(-1, 1024))


rep(torch_copy, "torch copy_ (%d OpenMP threads)" % torch.get_num_threads())
for nt in (1, 2, 4, 8, 16):
    pool = ThreadPoolExecutor(nt)

    def pooled():
        futs = []
        for i, b in enumerate(bags):
            src = b.reshape(-1, 1024).numpy()
            step = (8192 + nt - 1) // nt
            for r0 in range(0, 8192, step):
                futs.append(pool.submit(np.copyto, pv[i * 8192 + r0:i * 8192 + min(r0 + step, 8192)], src[r0:r0 + step]))
        for f in futs:
            f.result()
    rep(pooled, f"numpy copyto, pool of {nt}")
    pool.shutdown()
import os
print("cpus allowed", len(os.sched_getaffinity(0)), "cpu count", os.cpu_count())
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
