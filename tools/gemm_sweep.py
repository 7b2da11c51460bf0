#!/usr/bin/env python3
"""GPU micro-benchmark: every tile variant x split count of the fp32 MFMA GEMM on the shapes of the G+D step.
Back-to-back launches between two HIP events (queue stays full, no host gaps). Prints a table + JSON."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # (name, a_kc, b_kc, M, N, K)
    ("embedG_fwd", 1, 1, 8192, 384, 1024), ("embedGD_fwd", 1, 1, 8192, 512, 1024), ("embedD_fwd", 1, 1, 8192, 128, 1024),
    ("gates_fwd", 1, 1, 8192, 768, 384), ("dh_nn", 1, 0, 8192, 384, 768),
    ("dW1_tn", 0, 0, 384, 1024, 8192), ("dWGD_tn", 0, 0, 512, 1024, 8192), ("dWc_tn", 0, 0, 128, 1024, 8192),
    ("dWab_tn", 0, 0, 768, 384, 8192),
    ("embedG_fwd_32k", 1, 1, 32768, 384, 1024), ("dW1_tn_32k", 0, 0, 384, 1024, 32768),
]
TILES = [22, 23, 13, 12, 11]


def time_one(A, B, a_kc, b_kc, M, N, K, tile, splits, iters=20):
    out = torch.empty(M, N, device=DEV)
    for _ in range(3):
        ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, splits=splits, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, splits=splits, tile=tile)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    res = {}
    for name, a_kc, b_kc, M, N, K in SHAPES:
        A = torch.randn((M, K) if a_kc else (K, M), device=DEV)
        B = torch.randn((N, K) if b_kc else (K, N), device=DEV)
        flops = 2.0 * M * N * K
        rows = []
        split_opts = [1] if M * N >= 8192 * 128 else [4, 8, 11, 16, 22, 32]
        if name in ("embedD_fwd",):
            split_opts = [1, 2, 4]
        for tile in TILES:
            for sp in split_opts:
                us = time_one(A, B, a_kc, b_kc, M, N, K, tile, sp)
                tm, tn = tile // 10, tile % 10
                wgs = -(-M // (64 * tm)) * -(-N // (64 * tn)) * sp
                rows.append((us, tile, sp, wgs, flops / us / 1e6))
        rows.sort()
        res[name] = [{"us": round(r[0], 1), "tile": r[1], "splits": r[2], "wgs": r[3], "tflops": round(r[4], 1)} for r in rows]
        auto = time_one(A, B, a_kc, b_kc, M, N, K, 0, ops.auto_splits(M, N, K))
        print(f"{name:16s} M={M} N={N} K={K}  auto: {auto:7.1f} us ({flops / auto / 1e6:5.1f} TF)   best: " +
              "  ".join(f"t{r[1]}/s{r[2]}({r[3]}wg) {r[0]:.1f}us {r[4]:.0f}TF" for r in rows[:4]), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm_sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
