#!/usr/bin/env python3
"""Randomised check of the contraction engine (ops.gemm -> advmil_gemm_f32_tiled) against float64: random shapes (contiguous
dimensions multiples of 4; small, ragged, slab-sized), the four operand layouts, both arithmetic modes, random epilogue (bias, two
activations with a split column, rank-1 term with row segments, mask, accumulate, alpha), the plan's tile or a forced one, split-K, and
pre-split operand planes for A and / or B. usage: gemm_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
ACT = {0: lambda v: v, 1: torch.relu, 2: torch.tanh, 3: torch.sigmoid}
worst = {"exact": 0.0, "bf16x3": 0.0}
for case in range(ncase):
    mode = rnd.choice(("exact", "bf16x3"))
    ops.set_gemm_mode(mode)
    cls = rnd.random()
    if cls < 0.4:
        M, N, K = 4 * rnd.randint(1, 64), 4 * rnd.randint(1, 150), 4 * rnd.randint(1, 150)
    elif cls < 0.8:
        M, N, K = 4 * rnd.randint(64, 1500), 4 * rnd.randint(8, 200), 4 * rnd.randint(4, 300)
    else:
        M, N, K = 256 * rnd.randint(16, 160), 64 * rnd.randint(1, 12), 32 * rnd.randint(2, 32)
    a_kc, b_kc = rnd.random() < 0.6, rnd.random() < 0.6
    if rnd.random() < 0.25:                           # a weight-gradient shape: contraction over the long dimension
        M, K = K, M
        a_kc = b_kc = False
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    Ad = (A if a_kc else A.t().contiguous()).to(dev)
    Bd = (B.t().contiguous() if b_kc else B).to(dev)
    kw, alpha = {}, 1.0
    if rnd.random() < 0.5:
        alpha = rnd.choice((1.0, 0.5, -2.0)); kw["alpha"] = alpha
    pre = alpha * ref
    if rnd.random() < 0.5:
        bias = torch.randn(N, generator=g); kw["bias"] = bias.to(dev); pre = pre + bias.double()
    if rnd.random() < 0.3:
        nseg = rnd.randint(1, 5)
        rowv = torch.randn(M, generator=g); colv = torch.randn(nseg, N, generator=g)
        seg = torch.sort(torch.randint(0, nseg, (M,), generator=g)).values.to(torch.int32)
        kw.update(rowv=rowv.to(dev), colv=colv.to(dev), rowseg=seg.to(dev))
        pre = pre + rowv.double()[:, None] * colv.double()[seg.long()]
    a0, a1 = rnd.randint(0, 3), rnd.randint(0, 3)
    if rnd.random() < 0.5 and N >= 8:
        split = 4 * rnd.randint(1, N // 4 - 1) if N > 8 else 4
        kw.update(act0=a0, act1=a1, act_split=split)
        v = torch.cat([ACT[a0](pre[:, :split]), ACT[a1](pre[:, split:])], dim=1)
    else:
        kw.update(act0=a0)
        v = ACT[a0](pre)
    if rnd.random() < 0.25:
        mref = torch.randn(M, N, generator=g); sc = rnd.choice((1.0, 1.25))
        kw.update(maskref=mref.to(dev), mask_scale=sc)
        v = v * (mref.double() > 0) * sc
    out = None
    if rnd.random() < 0.35:
        C0 = torch.randn(M, N, generator=g); out = C0.clone().to(dev); kw.update(out=out, accumulate=True)
        v = v + C0.double()
    if rnd.random() < 0.3:
        kw["splits"] = rnd.randint(1, 5)
    if mode == "bf16x3" and rnd.random() < 0.5 and K % 8 == 0 and M % 8 == 0 and N % 8 == 0:
        if rnd.random() < 0.7:
            kw["a_planes"] = ops.split_planes(Ad)
        if rnd.random() < 0.7:
            kw["b_planes"] = ops.split_planes(Bd)
    try:
        C = ops.gemm(Ad, Bd, a_kc, b_kc, M, N, K, **kw)
    except Exception as exc:
        print(f"case {case}: {mode} M,N,K={M},{N},{K} a_kc={a_kc} b_kc={b_kc} {sorted(kw)}: RAISED {type(exc).__name__}: {exc}")
        sys.exit(1)
    torch.cuda.synchronize()
    err = float((C.cpu().double() - v).abs().max() / (v.abs().max() + 1e-30))
    # error model: |A||B| accumulates like sqrt(K) x eps x the product scale; relative to the OUTPUT's scale (act / cancellation can
    # make it small) -> scale the allowance by max|pre| / max|v|
    amp = float(pre.abs().max() / (v.abs().max() + 1e-30)) if float(v.abs().max()) > 0 else 1.0
    # (fp32 accumulation over K terms: the exact mode's error grows like sqrt(K) eps -- 6e-6 of the product scale at K = 32 512)
    tol = (3e-6 * max(1.0, (K / 1024.0) ** 0.5) if mode == "exact" else 4e-5) * max(1.0, amp)
    worst[mode] = max(worst[mode], err / max(1.0, amp))
    ok = err <= tol and bool(torch.isfinite(C).all())
    if not ok or case % 20 == 0:
        print(f"case {case}: {mode} M,N,K={M},{N},{K} a_kc={a_kc} b_kc={b_kc} {sorted(k for k in kw if k != 'out')}: err {err:.2e} (tol {tol:.1e})"
              f" {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
ops.set_gemm_mode("exact")
print("all ok;", ncase, "cases; worst scaled error", worst)
