#!/usr/bin/env python3
"""Per-workgroup timeline of the plane-fed contraction (see stamp_gemm.sh)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import _lib, ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
lib = ctypes.CDLL(os.environ["ADVMIL_HIP_LIB"])
lib.advmil_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, M, N, K, t in (("gates", 131072, 768, 384, 83), ("embedG", 131072, 384, 1024, 83), ("embedD", 131072, 128, 1024, 82)):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    pa, pb = ops.split_planes(A), ops.split_planes(B)
    for _ in range(3):
        ops.gemm(A, B, True, True, M, N, K, out=out, bias=bias, act0=1, a_planes=pa, b_planes=pb, tile=t, splits=1)
    torch.cuda.synchronize()
    nwg = (M // 256) * (N // (64 * (t - 80)))
    n = min(nwg, 4096)
    buf = np.zeros(4 * 4096, dtype=np.uint64)
    rc = lib.advmil_debug_stamps(buf.ctypes.data, 4 * 4096)
    st = buf.reshape(4096, 4)[:n].astype(np.float64) / 100.0      # us
    t0 = st[:, 0].min()
    st -= t0
    d = np.diff(st, axis=1)
    print(f"{name} [{M},{N},{K}] tile {t}: {nwg} WGs (first {n} stamped) rc={rc}; kernel span {st[:, 3].max():.1f} us")
    print("   per-tile us (median / p90): until the first chunk is consumed %.2f / %.2f   rest of the K loop %.2f / %.2f   epilogue (issue) %.2f / %.2f   total %.2f / %.2f" % (
        np.median(d[:, 0]), np.percentile(d[:, 0], 90), np.median(d[:, 1]), np.percentile(d[:, 1], 90),
        np.median(d[:, 2]), np.percentile(d[:, 2], 90), np.median(st[:, 3] - st[:, 0]), np.percentile(st[:, 3] - st[:, 0], 90)))
    order = np.argsort(st[:, 0])
    starts = st[order, 0]
    print("   tile start times (us), every 256th in start order:", np.round(starts[::256], 1))
    nxt = np.sort(st[:, 0])
    # gap between a tile's epilogue end and the start of the next tile on the same workgroup (tile ids 256 apart)
    if n > 256:
        gaps = st[256:n, 0] - st[:n - 256, 3]
        print("   gap epilogue end -> next tile start on the same workgroup (us): median %.2f p90 %.2f" % (np.median(gaps), np.percentile(gaps, 90)))
    # concurrency: WGs per CU in sequence -> gaps between a WG's end and the next start on the chip
    print("   sum of per-WG totals / 256 CUs = %.1f us" % ((st[:, 3] - st[:, 0]).sum() / 256.0 * (nwg / n)))
