#!/usr/bin/env python3
"""Times the plane-fed NT contraction on the step's shapes (see ablate_gemm.sh); results of the ablated variants are garbage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, M, N, K, t in (("gates", 131072, 768, 384, 83), ("embedG", 131072, 384, 1024, 83), ("embedD", 131072, 128, 1024, 82), ("sq4096", 4032, 4032, 4096, 83)):
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    pa, pb = ops.split_planes(A), ops.split_planes(B)
    us = bench(lambda: ops.gemm(A, B, True, True, M, N, K, out=out, bias=bias, act0=1, a_planes=pa, b_planes=pb, tile=t, splits=1))
    print(f"  {name} [{M},{N},{K}] tile {t}: {us:.0f} us  {2.0 * M * N * K / us / 1e6:.0f} TF ({2.0 * M * N * K / us / 1e6 / 833.3:.2f} of roof)", flush=True)
