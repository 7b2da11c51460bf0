#!/usr/bin/env python3
"""dW1 = dpre^T X (TN over the slab rows) with dpre taken from planes written by the activation backward vs split on the fly."""
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
R, K = 131072, 1024
X = torch.randn(R, K, device=dev); px = ops.split_planes(X)
for N in (384, 192, 512, 128):
    dy = torch.randn(R, N, device=dev); y = torch.relu(torch.randn(R, N, device=dev))
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    for p in (0.0, 0.25):
        dpre, _ = ops.act_dropout_bwd(dy, y, ops.ACT_RELU, R, N, p, seed if p else None, 3, want_bias=False)
        pl = ops.Planes(torch.empty(R, N, dtype=torch.bfloat16, device=dev), torch.empty(R, N, dtype=torch.bfloat16, device=dev))
        none, _ = ops.act_dropout_bwd(dy, y, ops.ACT_RELU, R, N, p, seed if p else None, 3, want_bias=False, planes=pl, planes_only=True)
        ref = ops.split_planes(dpre)
        print(N, p, "planes of the backward == split_planes(dpre):", torch.equal(ref.hi, pl.hi), torch.equal(ref.lo, pl.lo), none is None)
        w0 = ops.gemm(dpre, X, False, False, N, K, R, b_planes=px)
        w1 = ops.gemm(None, X, False, False, N, K, R, a_planes=pl, b_planes=px)
        w2 = ops.gemm(dpre, X, False, False, N, K, R)
        print("   plan", ops.gemm_plan(N, K, R, False, False), "bit-identical (B planes / none):", torch.equal(w0, w1), torch.equal(w2, w1),
              "max diff", float((w0 - w1).abs().max()), "scale", float(w0.abs().max()),
              f"fly {bench(lambda: ops.gemm(dpre, X, False, False, N, K, R, b_planes=px)):.0f} us planes {bench(lambda: ops.gemm(None, X, False, False, N, K, R, a_planes=pl, b_planes=px)):.0f} us")
