#!/usr/bin/env python3
"""In-graph cost of the smallest launches: a trivial kernel vs the 64x64-tile contraction on [B, d]-sized operands vs torch.mm."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402
ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
def bench(fn, n=20, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)
t8 = torch.randn(64, device=dev); pl = ops.split_planes(t8)
print(f"split_planes of 64 floats (trivial kernel): {bench(lambda: ops.split_planes(t8, pl)):.2f} us")
z = torch.zeros(16, device=dev)
print(f"aten fill of 16 floats: {bench(lambda: z.fill_(1.0)):.2f} us")
for M, N, K in ((16, 64, 128), (16, 384, 384), (16, 384, 16), (32, 128, 64)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
    t0 = bench(lambda: ops.gemm(A, W, True, True, M, N, K, out=out, bias=b, act0=1))
    t1 = bench(lambda: torch.addmm(b, A, W.t(), out=out))
    print(f"[{M},{N},{K}] NT: ours {t0:.2f} us   torch.addmm {t1:.2f} us")
