#!/usr/bin/env python3
"""Launch ONE GEMM shape N times (for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / SQ passes, run separately).
usage: pmc_gemm.py M N K a_kc b_kc [iters] [planes] [skew]   (planes = 1: operands arrive as bf16 planes -> the plane-fed LDS-DMA
kernel; skew >= 0: the hi and lo plane of every A operand live in ONE allocation, lo starting `skew` bytes behind the end of hi --
placement experiment for the bimodal HBM fetch of the plane-fed kernel; default: two separate allocations)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

M, N, K, a_kc, b_kc = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
planes = len(sys.argv) > 7 and sys.argv[7] == "1"
dev = "cuda:0"
# a pool of distinct operands larger than the 256 MB Infinity Cache so that reads come from HBM
npool = max(2, int(400e6 // (4 * M * K)) + 1)
As = [torch.randn((M, K) if a_kc else (K, M), device=dev) for _ in range(npool)]
B = torch.randn((N, K) if b_kc else (K, N), device=dev)
out = torch.empty(M, N, device=dev)
skew = int(sys.argv[8]) if len(sys.argv) > 8 else -1
KEEP = []


def planes_of(a):
    if skew < 0:
        return ops.split_planes(a)
    if skew == 1:                           # row-interleaved: [row][hi K | lo K], one allocation, row pitch 2K
        d = ops.split_planes(a)
        buf = torch.empty(a.shape[0], 2 * a.shape[1], dtype=torch.bfloat16, device=dev)
        buf[:, :a.shape[1]].copy_(d.hi); buf[:, a.shape[1]:].copy_(d.lo)
        KEEP.append(buf)
        return ops.Planes(buf[:, :a.shape[1]], buf[:, a.shape[1]:])
    n = a.numel()
    buf = torch.empty(2 * n + skew // 2 + 64, dtype=torch.bfloat16, device=dev)
    out = ops.Planes(buf[:n].view(a.shape), buf[n + skew // 2:2 * n + skew // 2].view(a.shape))
    KEEP.append(buf)
    return ops.split_planes(a, out=out)


pas = [planes_of(a) for a in As] if planes else None
pb = ops.split_planes(B) if planes else None
for i in range(iters):
    if planes:
        ops.gemm(None if skew == 1 else As[i % npool], B, bool(a_kc), bool(b_kc), M, N, K, out=out, a_planes=pas[i % npool], b_planes=pb)
    else:
        ops.gemm(As[i % npool], B, bool(a_kc), bool(b_kc), M, N, K, out=out)
torch.cuda.synchronize()
print("done", M, N, K, ops.gemm_plan(M, N, K))
