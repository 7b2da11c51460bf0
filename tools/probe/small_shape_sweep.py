#!/usr/bin/env python3
"""Tile sweep of the contraction engine on the step's LAUNCH-BOUND shapes (region- and bag-level layers: M = 16 .. 16384 rows, N, K <= 384),
back to back between two HIP events, both arithmetic modes: is the plan's 64x64 tile the right choice there, and what is the floor?
usage: small_shape_sweep.py [mode=bf16x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_gemm_mode(mode)
DEV = "cuda:0"
SHAPES = [(1, 1, 16384, 256, 128), (1, 1, 16384, 128, 256), (1, 1, 16384, 64, 128), (1, 1, 16384, 128, 64), (1, 0, 16384, 128, 256),
          (1, 0, 16384, 64, 128), (1, 1, 8192, 256, 128), (1, 1, 8192, 64, 128), (1, 1, 2048, 256, 128), (1, 1, 2048, 64, 128),
          (1, 1, 1024, 128, 64), (1, 1, 16, 384, 384), (1, 0, 16, 384, 384), (1, 1, 32, 128, 64)]


def time_one(A, B, a_kc, b_kc, M, N, K, tile, iters=50):
    out = torch.empty(M, N, device=DEV)
    try:
        for _ in range(3):
            ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, splits=1, tile=tile)
    except Exception:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, splits=1, tile=tile)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for a_kc, b_kc, M, N, K in SHAPES:
    A = torch.randn((M, K) if a_kc else (K, M), device=DEV)
    B = torch.randn((N, K) if b_kc else (K, N), device=DEV)
    res = []
    for tile in (0, 11, 12, 13, 22, 23):
        us = time_one(A, B, a_kc, b_kc, M, N, K, tile)
        if us is not None:
            res.append((tile, us))
    print(f"{mode} a_kc {a_kc} b_kc {b_kc} M {M:6d} N {N:4d} K {K:4d}: " + "  ".join(f"t{t}: {u:5.1f} us" for t, u in res) +
          f"   plan tile {ops.gemm_plan(M, N, K, bool(a_kc), bool(b_kc))}", flush=True)
