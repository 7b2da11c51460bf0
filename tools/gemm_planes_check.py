#!/usr/bin/env python3
"""bf16x3 mode: the slab-sized NT contractions with operands split on the fly (generic kernel) vs taken from pre-split planes
through the plane-fed LDS-DMA kernel (tile 82/83), incl. bit-identity of the results and the fused gate-score mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
SHAPES = [("embedG", 131072, 384, 1024), ("gates", 131072, 768, 384), ("embedD", 131072, 128, 1024), ("dh_nt", 131072, 384, 768),
          ("embedG32k", 524288, 384, 1024), ("small", 8192, 384, 1024), ("embedGD", 131072, 512, 1024)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    out0 = torch.empty(M, N, device=dev)
    out1 = torch.empty(M, N, device=dev)
    pa, pb = ops.split_planes(A), ops.split_planes(B)
    ops.gemm(A, B, True, True, M, N, K, out=out0, bias=bias, act0=1)
    row = [name, f"plan={ops.gemm_plan(M, N, K)} planes_tile={ops.gemm_plan_planes(M, N, K)}"]
    us0 = bench(lambda: ops.gemm(A, B, True, True, M, N, K, out=out0, bias=bias, act0=1))
    row.append(f"fly {us0:.0f}us {2.0 * M * N * K / us0 / 1e6:.0f}TF")
    for t in (82, 83, 85):
        if N % (64 * (4 if t == 85 else t % 10)):
            continue
        ops.gemm(A, B, True, True, M, N, K, out=out1, bias=bias, act0=1, a_planes=pa, b_planes=pb, tile=t, splits=1)
        same = torch.equal(out0, out1)
        us = bench(lambda: ops.gemm(A, B, True, True, M, N, K, out=out1, bias=bias, act0=1, a_planes=pa, b_planes=pb, tile=t, splits=1))
        row.append(f"t{t} {us:.0f}us {2.0 * M * N * K / us / 1e6:.0f}TF ({2.0 * M * N * K / us / 1e6 / 833.3:.2f}) bit-identical={same}")
    print("  ".join(row), flush=True)
    del A, B, out0, out1, pa, pb
# fused gate score through the plane kernel
M, D = 131072, 384
h = torch.randn(M, D, device=dev); Wi = torch.randn(2 * D, D, device=dev) * 0.05; bi = torch.randn(2 * D, device=dev) * 0.1
wc = torch.randn(D, device=dev)
s0 = ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc).sum(dim=1)
ph, pw = ops.split_planes(h), ops.split_planes(Wi)
s1 = ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc, a_planes=ph, b_planes=pw).sum(dim=1)
print("gate-score planes vs fly max abs diff", float((s0 - s1).abs().max()), "ref scale", float(s0.abs().max()))
us0 = bench(lambda: ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc))
us1 = bench(lambda: ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc, a_planes=ph, b_planes=pw))
print(f"gate-score: fly {us0:.0f} us, planes {us1:.0f} us = {2.0 * M * 2 * D * D / us1 / 1e6:.0f} TF")
us = bench(lambda: ops.split_planes(h, ph))
print(f"split_planes of {h.numel() * 4 / 1e6:.0f} MB: {us:.0f} us")
