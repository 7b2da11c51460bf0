#!/usr/bin/env python3
"""Latency of the [B, d]-sized contractions of the heads/tails (launch-bound shapes), both arithmetic modes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

dev = "cuda:0"
SHAPES = [("head 16x192x768 NT", 16, 192, 768, 1, 1), ("rho 16x384x384 NT", 16, 384, 384, 1, 1), ("fc1 8192x64x128 NT", 8192, 64, 128, 1, 1),
          ("fc1b 8192x128x64 NT", 8192, 128, 64, 1, 1), ("gap 8192x256x128 NT", 8192, 256, 128, 1, 1), ("dW 64x128x8192 TN", 64, 128, 8192, 0, 0),
          ("dW 256x128x8192 TN", 256, 128, 8192, 0, 0), ("dW 192x768x16 TN", 192, 768, 16, 0, 0), ("dx 8192x128x256 NN", 8192, 128, 256, 1, 0),
          ("dx 16x768x192 NN", 16, 768, 192, 1, 0), ("1bag fc 512x64x128 NT", 512, 64, 128, 1, 1), ("1bag dW 64x128x512 TN", 64, 128, 512, 0, 0)]


def bench(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 20 * 20)


for mode in ("exact", "bf16x3"):
    ops.set_gemm_mode(mode)
    for name, M, N, K, a_kc, b_kc in SHAPES:
        A = torch.randn((M, K) if a_kc else (K, M), device=dev)
        B = torch.randn((N, K) if b_kc else (K, N), device=dev)
        out = torch.empty(M, N, device=dev)
        us = bench(lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out))
        tref = bench(lambda: torch.mm(A if a_kc else A.t(), B.t() if b_kc else B, out=out))
        print(f"{mode:7s} {name:26s} plan={ops.gemm_plan(M, N, K)}  ours {us:6.1f} us   torch.mm {tref:6.1f} us")
