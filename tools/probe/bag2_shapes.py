#!/usr/bin/env python3
"""The slab contractions of the 2-bag step (16384 rows) and of the 1-bag step (8192): tile / split / kernel variants, in-graph time."""
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
        return f"{label}:ERR({str(ex)[:40]})"


for R in (8192, 16384):
    X = torch.randn(R, 1024, device=dev)
    Xp = ops.split_planes(X)
    W1 = torch.randn(384, 1024, device=dev) * 0.03
    W1p = ops.split_planes(W1)
    WD = torch.randn(128, 1024, device=dev) * 0.03
    WDp = ops.split_planes(WD)
    b1 = torch.randn(384, device=dev)
    # FC1: NT R x 384 x 1024, relu epilogue
    row = [f"FC1 [{R},384,1024]"]
    for t in (23, 22, 12):
        row.append(tryb(f"t{t}", lambda: ops.gemm(X, W1, True, True, R, 384, 1024, bias=b1, act0=1, tile=t)))
    for t in (22, 12):
        row.append(tryb(f"t{t}+pl", lambda: ops.gemm(X, W1, True, True, R, 384, 1024, bias=b1, act0=1, tile=t, a_planes=Xp, b_planes=W1p)))
    for t in (82, 83):
        row.append(tryb(f"t{t}", lambda: ops.gemm(X, W1, True, True, R, 384, 1024, bias=b1, act0=1, tile=t, a_planes=Xp, b_planes=W1p)))
    print("  ".join(row), flush=True)
    row = [f"DFC [{R},128,1024]"]
    for t in (11, 12, 22):
        for sp in (1, 2, 4):
            row.append(tryb(f"t{t}/s{sp}", lambda: ops.gemm(X, WD, True, True, R, 128, 1024, tile=t, splits=sp)))
    for t in (11, 12, 22):
        row.append(tryb(f"t{t}+pl", lambda: ops.gemm(X, WD, True, True, R, 128, 1024, tile=t, splits=1, a_planes=Xp, b_planes=WDp)))
    row.append(tryb("t82", lambda: ops.gemm(X, WD, True, True, R, 128, 1024, tile=82, a_planes=Xp, b_planes=WDp)))
    print("  ".join(row), flush=True)
    # dW_D: TN 128 x 1024 x R, both planes
    dY = torch.randn(R, 128, device=dev)
    dYp = ops.split_planes(dY)
    out = torch.zeros(128, 1024, device=dev)
    row = [f"dWD [128,1024,{R}] plan={ops.gemm_plan(128, 1024, R, False, False)} tn={ops.gemm_plan_tn_planes(128, 1024, R)}"]
    for t, sps in ((91, (8, 16, 32, 64)), (24, (8, 16, 32, 64)), (22, (8, 16, 32))):
        for sp in sps:
            if R // sp < 128:
                continue
            row.append(tryb(f"t{t}/s{sp}", lambda: ops.gemm(None, X, False, False, 128, 1024, R, out=out, ldc=1024, accumulate=True, tile=t, splits=sp,
                                                          a_planes=dYp, b_planes=Xp)))
    print("  ".join(row), flush=True)
    # dW1: TN 384 x 1024 x R
    dP = torch.randn(R, 384, device=dev)
    dPp = ops.split_planes(dP)
    out1 = torch.zeros(384, 1024, device=dev)
    row = [f"dW1 [384,1024,{R}] plan={ops.gemm_plan(384, 1024, R, False, False)} tn={ops.gemm_plan_tn_planes(384, 1024, R)}"]
    for t, sps in ((91, (8, 16, 20, 32)), (34, (8, 16, 32))):
        for sp in sps:
            if R // sp < 128:
                continue
            row.append(tryb(f"t{t}/s{sp}", lambda: ops.gemm(None, X, False, False, 384, 1024, R, out=out1, ldc=1024, accumulate=True, tile=t, splits=sp,
                                                          a_planes=dPp, b_planes=Xp)))
    print("  ".join(row), flush=True)
    # dWab: TN 768 x 384 x R
    dG = torch.randn(R, 768, device=dev)
    dGp = ops.split_planes(dG)
    h = torch.randn(R, 384, device=dev)
    hp = ops.split_planes(h)
    out2 = torch.zeros(768, 384, device=dev)
    row = [f"dWab [768,384,{R}] plan={ops.gemm_plan(768, 384, R, False, False)} tn={ops.gemm_plan_tn_planes(768, 384, R)}"]
    for t, sps in ((92, (8, 16, 26, 32)), (43, (8, 16, 32))):
        for sp in sps:
            if R // sp < 128:
                continue
            row.append(tryb(f"t{t}/s{sp}", lambda: ops.gemm(None, h, False, False, 768, 384, R, out=out2, ldc=384, accumulate=True, tile=t, splits=sp,
                                                          a_planes=dGp, b_planes=hp)))
    print("  ".join(row), flush=True)
    # dh: NN R x 384 x 768, A = dG planes only
    Wab = torch.randn(768, 384, device=dev) * 0.05
    row = [f"dh [{R},384,768]"]
    for t in (43, 22, 12):
        row.append(tryb(f"t{t}", lambda: ops.gemm(None, Wab, True, False, R, 384, 768, tile=t, a_planes=dGp)))
    print("  ".join(row), flush=True)
    # gates: NT R x 768 x 384 storing, planes
    Wg = torch.randn(768, 384, device=dev) * 0.05
    Wgp = ops.split_planes(Wg)
    row = [f"gates [{R},768,384]"]
    for t in (83, 82, 85, 23, 22):
        row.append(tryb(f"t{t}", lambda: ops.gemm(h, Wg, True, True, R, 768, 384, tile=t, a_planes=hp, b_planes=Wgp)))
    print("  ".join(row), flush=True)
