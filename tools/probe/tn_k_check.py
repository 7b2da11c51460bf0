#!/usr/bin/env python3
"""TN contraction dW = A^T B over K rows that are not a multiple of the usual tile sizes (found through pad_fuzz seed 105: 10592 rows):
the planned launch against float64, both arithmetic modes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
for mode in ("exact", "bf16x3"):
    ops.set_gemm_mode(mode)
    for (M, N) in ((384, 1024), (128, 1024), (768, 384)):
        for K in (10592, 10560, 10624, 10592 + 16, 6448, 19872, 4112):
            A = torch.randn(K, M, generator=g).to(dev)
            B = torch.randn(K, N, generator=g).to(dev)
            ref = (A.double().t() @ B.double())
            out = ops.gemm(A, B, False, False, M, N, K)
            err = (out.double() - ref).abs()
            bad = int((err > 1e-4 * float(ref.abs().max())).sum())
            print(f"{mode} [{M},{N},{K}] plan {ops.gemm_plan(M, N, K, False, False)}: max err {float(err.max() / ref.abs().max()):.2e}  entries off {bad}"
                  + ("" if bad == 0 else f"  rows {sorted(set((err > 1e-4 * float(ref.abs().max())).nonzero()[:, 0].tolist()))[:12]}"), flush=True)
