#!/usr/bin/env python3
"""bf16x3 mode: time the slab-sized contractions with operands split on the fly vs taken from pre-split planes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
SHAPES = [("embedG", 131072, 384, 1024, 1, 1), ("gates", 131072, 768, 384, 1, 1), ("embedD", 131072, 128, 1024, 1, 1),
          ("dh", 131072, 384, 768, 1, 0), ("dW1", 384, 1024, 131072, 0, 0), ("dWab", 768, 384, 131072, 0, 0)]


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


for name, M, N, K, a_kc, b_kc in SHAPES:
    A = torch.randn((M, K) if a_kc else (K, M), device=dev)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    pa, pb = ops.split_planes(A), ops.split_planes(B)
    cp = ops.Planes.empty_like(out)
    row = [name, str(ops.gemm_plan(M, N, K))]
    for label, kw in [("fly", {}), ("B", dict(b_planes=pb)), ("A", dict(a_planes=pa)), ("A+B", dict(a_planes=pa, b_planes=pb)),
                      ("A+B+emitC", dict(a_planes=pa, b_planes=pb, c_planes=cp))]:
        us = bench(lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, **kw))
        row.append(f"{label} {us:.0f}us {2.0 * M * N * K / us / 1e6:.0f}TF")
    print("  ".join(row))
us = bench(lambda: ops.split_planes(A, pa))
print(f"split_planes of {A.numel() * 4 / 1e6:.0f} MB: {us:.0f} us")
