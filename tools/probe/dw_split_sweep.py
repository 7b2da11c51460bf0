#!/usr/bin/env python3
"""Split-count sweep of the slab-sized weight-gradient contractions on their 8-wave tiles, operands as the step passes them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
R = 131072
for name, M, N, tile, pa, pb in (("dW1", 384, 1024, 34, True, True), ("dWab", 768, 384, 43, True, False), ("dWD", 128, 1024, 24, True, True)):
    A = torch.randn(R, M, device=dev); B = torch.randn(R, N, device=dev)
    apl = ops.split_planes(A) if pa else None; bpl = ops.split_planes(B) if pb else None
    row = [f"{name} [{M},{N},{R}] plan={ops.gemm_plan(M, N, R, False, False)}"]
    for sp in (8, 12, 16, 21, 24, 32, 42, 48, 64, 85, 128):
        try:
            t = bench(lambda: ops.gemm(None if pa else A, B, False, False, M, N, R, tile=tile, splits=sp, a_planes=apl, b_planes=bpl))
        except Exception as e:
            row.append(f"s{sp}:ERR"); continue
        row.append(f"s{sp}:{t:.0f}us")
    print("  ".join(row), flush=True)
