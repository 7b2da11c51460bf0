#!/usr/bin/env python3
"""Weight-gradient contractions dY^T X over the slab rows: B operand (X) on the fly vs from its resident planes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402
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
for name, M, N, K in (("dW1", 384, 1024, 131072), ("dWD", 128, 1024, 131072), ("dWab", 768, 384, 131072)):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    pb = ops.split_planes(B)
    o0 = ops.gemm(A, B, False, False, M, N, K)
    o1 = ops.gemm(A, B, False, False, M, N, K, b_planes=pb)
    us0 = bench(lambda: ops.gemm(A, B, False, False, M, N, K))
    us1 = bench(lambda: ops.gemm(A, B, False, False, M, N, K, b_planes=pb))
    print(f"{name} [{M},{N},{K}] plan {ops.gemm_plan(M, N, K, False, False)}: fly {us0:.0f} us ({2.0*M*N*K/us0/1e6/833.3:.2f})  X planes {us1:.0f} us ({2.0*M*N*K/us1/1e6/833.3:.2f})  bit-identical={torch.equal(o0, o1)}")
