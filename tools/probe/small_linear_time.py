#!/usr/bin/env python3
"""Dev probe: in-graph time of one [B, d] linear layer forward + backward on the small_linear kernels vs the 64x64-tile contraction path.
usage: small_linear_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
for (M, N, K) in ((1, 384, 384), (2, 384, 384), (16, 384, 384), (16, 192, 768), (16, 128, 64), (32, 64, 128)):
    res = {}
    for small in (True, False):
        ops.SMALL_LINEAR = small
        x = torch.randn(M, K, device=dev, requires_grad=True)
        W = torch.randn(N, K, device=dev, requires_grad=True)
        b = torch.randn(N, device=dev, requires_grad=True)
        go = torch.randn(M, N, device=dev)
        rng = ops.DeviceRng(dev, seed=1)

        def fb():
            y = ops.linear_act(x, W, b, "relu", 0.25, rng, "t")
            y.backward(go)
        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fb()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res[small] = e0.elapsed_time(e1) * 1e3 / 400
    ops.SMALL_LINEAR = True
    print(f"[{M} x {K}] -> {N}: fwd+bwd in-graph  small_linear {res[True]:.1f} us   contraction path {res[False]:.1f} us", flush=True)
