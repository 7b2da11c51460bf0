#!/usr/bin/env python3
"""Back-to-back timing of the bf16 NT engine on the generator's slab shapes (tile x splits)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops
DEV = "cuda:0"
SHAPES = [("embedG_fwd", 131072, 384, 1024), ("gates_fwd", 131072, 768, 384), ("dh", 131072, 384, 768),
          ("dW1", 384, 1024, 131072), ("dWab", 768, 384, 131072), ("embedG_8k", 8192, 384, 1024), ("dW1_8k", 384, 1024, 8192)]
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device=DEV).bfloat16(); B = torch.randn(N, K, device=DEV).bfloat16()
    out = torch.empty(M, N, device=DEV)
    rows = []
    for tile in (22, 12, 11):
        for sp in ([1] if M * N >= 8192 * 128 else [4, 8, 16, 32, 64]):
            if sp > K // 64:
                continue
            for _ in range(2):
                ops.gemm_bf16(A, B, M, N, K, out=out, splits=sp, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_bf16(A, B, M, N, K, out=out, splits=sp, tile=tile)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            rows.append((us, tile, sp, 2.0 * M * N * K / us / 1e6))
    rows.sort()
    pt, ps = ops.gemm_bf16_plan(M, N, K)
    print(f"{name:12s} {M}x{N}x{K} plan=t{pt}/s{ps}  " + "  ".join(f"t{r[1]}/s{r[2]} {r[0]:.0f}us {r[3]:.0f}TF" for r in rows[:5]), flush=True)
