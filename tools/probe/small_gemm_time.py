#!/usr/bin/env python3
"""In-graph time of the launch-bound 64x64-tile contractions of a step ([B, d] and region-level layers, split-K weight gradients)."""
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
tot = 0.0
for a_kc, b_kc, M, N, K in ((1, 1, 16, 384, 384), (1, 1, 16, 192, 384), (1, 1, 32, 128, 64), (1, 1, 16, 128, 64), (1, 1, 16, 64, 128),
                            (1, 0, 16, 384, 384), (1, 0, 32, 64, 128), (1, 0, 16, 384, 192), (0, 0, 384, 384, 16), (0, 0, 128, 64, 32),
                            (0, 0, 256, 128, 16384), (0, 0, 128, 64, 16384), (1, 1, 16384, 64, 128), (1, 1, 8192, 256, 128), (1, 1, 16, 1024, 1024)):
    A = torch.randn((M, K) if a_kc else (K, M), device=dev); B = torch.randn((N, K) if b_kc else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    tile, sp = ops.gemm_plan(M, N, K, bool(a_kc), bool(b_kc))
    t = bench(lambda: ops.gemm(A, B, bool(a_kc), bool(b_kc), M, N, K, out=out))
    ref = (A if a_kc else A.t()).double() @ (B.t() if b_kc else B).double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    tot += t
    print(f"a_kc={a_kc} b_kc={b_kc} [{M},{N},{K}] tile {tile} splits {sp}: {t:.2f} us  relerr {err:.1e}  csum {float(out.double().sum()):.10e}")
print(f"total {tot:.1f} us")
