#!/usr/bin/env python3
"""Back-to-back timing of the fp32 engine on the 16x8k slab shapes for every tile (and a few splits for the TN forms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops
for (name, M, N, K, akc, bkc) in (("embedG", 131072, 384, 1024, 1, 1), ("gates", 131072, 768, 384, 1, 1), ("embedD", 131072, 128, 1024, 1, 1),
                                  ("dh", 131072, 384, 768, 1, 0), ("dW1", 384, 1024, 131072, 0, 0), ("dWab", 768, 384, 131072, 0, 0)):
    A = torch.randn((M, K) if akc else (K, M), device="cuda"); B = torch.randn((N, K) if bkc else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    rows = []
    for tile in ((43, 42, 23, 22, 13, 12, 11) if ops.get_gemm_mode() == "bf16x3" else (23, 22, 13, 12, 11)):
        for sp in ([1] if M > 4096 else [16, 32, 64]):
            for _ in range(2): ops.gemm(A, B, akc, bkc, M, N, K, out=out, splits=sp, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(A, B, akc, bkc, M, N, K, out=out, splits=sp, tile=tile)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            rows.append((us, tile, sp))
    rows.sort()
    print(f"{name:7s} plan={ops.gemm_plan(M, N, K, akc, bkc)} " + "  ".join(f"t{t}/s{sp} {us:.0f}us {2.0*M*N*K/us/1e6:.0f}TF" for us, t, sp in rows[:6]), flush=True)
