#!/usr/bin/env python3
"""Time the deep-K weight-gradient contractions of the 16 x 8192 ABMIL step on the plane-fed TN kernel against the generic kernel
(both operands as planes in both cases). usage: tn_planes_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K in ((384, 1024, 131072), (768, 384, 131072), (128, 1024, 131072)):
    As = [torch.randn(K, M, device=dev) for _ in range(2)]
    Bs = [torch.randn(K, N, device=dev) for _ in range(2)]
    pas, pbs = [ops.split_planes(a) for a in As], [ops.split_planes(b) for b in Bs]
    out = torch.zeros(M, N, device=dev)
    k = [0]

    def run(**kw):
        i = k[0] % 2
        k[0] += 1
        ops.gemm(As[i], Bs[i], False, False, M, N, K, out=out, ldc=N, accumulate=True, a_planes=pas[i], b_planes=pbs[i], **kw)

    gt, gs = ops.gemm_plan(M, N, K, False, False)
    tt, ts = ops.gemm_plan_tn_planes(M, N, K)
    us_g = timed(lambda: run(tile=gt, splits=gs))
    us_t = timed(lambda: run()) if tt else float("nan")
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: generic tile {gt} x {gs} splits {us_g:7.1f} us = {fl / us_g / 1e6:6.1f} TF ({fl / us_g / 1e6 / 833.3:.3f});  "
          f"plane-fed TN tile {tt} x {ts} splits {us_t:7.1f} us = {fl / us_t / 1e6:6.1f} TF ({fl / us_t / 1e6 / 833.3:.3f})")
    for sp in (8, 12, 16, 20, 24, 32, 40):
        if tt and K // sp >= 1024:
            us = timed(lambda: run(tile=tt, splits=sp), 10)
            print(f"      tile {tt} splits {sp:3d}: {us:7.1f} us")
