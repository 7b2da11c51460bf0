#!/usr/bin/env python3
"""Deep-K weight-gradient contractions dY^T X on region-level shapes: tile / split-count sweep (bf16x3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for M, N, K in ((256, 128, 16384), (384, 384, 4096)):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    ref = None
    row = [f"[{M},{N},{K}] plan={ops.gemm_plan(M, N, K, False, False)}"]
    for tile in (11, 12, 22, 23):
        for sp in (8, 16, 22, 32, 43, 64):
            if K // sp < 256: continue
            out = ops.gemm(A, B, False, False, M, N, K, tile=tile, splits=sp)
            if ref is None: ref = A.double().t() @ B.double()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            t = bench(lambda: ops.gemm(A, B, False, False, M, N, K, tile=tile, splits=sp))
            row.append(f"t{tile}/s{sp}:{t:.0f}us" + ("" if err < 2e-5 else f"(ERR {err:.1e})"))
    print("  ".join(row), flush=True)
