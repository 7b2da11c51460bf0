#!/usr/bin/env python3
"""The three weight-gradient contractions behind the fused region network (dWab 256x128, dW2 128x64, dW1 64x128 over K = region rows):
tile / split sweep at the K of the 1-bag, 16-bag ABMIL and ESAT 32k steps, accumulate into a destination as the step does (bf16x3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev = "cuda:0"


def bench(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for K in (512, 8192, 32768):
    for M, N in ((256, 128), (128, 64), (64, 128)):
        A = torch.randn(K, M, device=dev)
        B = torch.randn(K, N, device=dev)
        out = torch.zeros(M, N, device=dev)
        row = [f"[{M},{N},{K}] plan={ops.gemm_plan(M, N, K, False, False)}"]
        t = bench(lambda: ops.gemm(A, B, False, False, M, N, K, out=out, ldc=N, accumulate=True))
        row.append(f"plan:{t:.1f}us")
        for tile in (11, 12, 22):
            for sp in (1, 4, 16, 32, 64, 128, 256):
                if K // sp < 64 or (K // sp) % 32:
                    continue
                try:
                    t = bench(lambda: ops.gemm(A, B, False, False, M, N, K, out=out, ldc=N, accumulate=True, tile=tile, splits=sp))
                except Exception as ex:
                    row.append(f"t{tile}/s{sp}:ERR")
                    continue
                row.append(f"t{tile}/s{sp}:{t:.1f}")
        print("  ".join(row), flush=True)
