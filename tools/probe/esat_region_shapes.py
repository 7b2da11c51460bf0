#!/usr/bin/env python3
"""ESAT 32k's region-level contractions (32768 rows x {384, 768, 1152} x K 384 / 768): generic tiles with and without pre-split operands against
the plane-fed kernel, in-graph time. Is there anything in handing these layers operand planes (ops.ROW_PLANES)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev = "cuda:0"


def bench(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def tryb(label, fn):
    try:
        return f"{label}:{bench(fn):.1f}"
    except Exception as ex:
        return f"{label}:ERR({str(ex)[:30]})"


R = 32768
for (N, K) in ((384, 384), (768, 384), (1152, 384), (384, 768)):
    xs = [torch.randn(R, K, device=dev) for _ in range(3)]          # rotate inputs: three slabs > L2
    xp = [ops.split_planes(x) for x in xs]
    W = torch.randn(N, K, device=dev) * 0.05
    Wp = ops.split_planes(W)
    b = torch.randn(N, device=dev)
    k = [0]

    def run(tile, planes):
        i = k[0] % 3
        k[0] += 1
        if planes:
            ops.gemm(xs[i], W, True, True, R, N, K, bias=b, act0=1, tile=tile, a_planes=xp[i], b_planes=Wp)
        else:
            ops.gemm(xs[i], W, True, True, R, N, K, bias=b, act0=1, tile=tile)
    row = [f"NT [{R},{N},{K}] plan {ops.gemm_plan(R, N, K, True, True)} planes-plan {ops.gemm_plan_planes(R, N, K)}"]
    for t in (23, 22, 12):
        row.append(tryb(f"t{t}", lambda: run(t, False)))
    for t in (22, 12, 82, 83):
        row.append(tryb(f"t{t}+pl", lambda: run(t, True)))
    print("  ".join(row), flush=True)
