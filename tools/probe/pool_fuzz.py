#!/usr/bin/env python3
"""Randomised check of the segmented softmax-pooling call (ops.softmax_pool / softmax_pool_bwd -> advmil_softmax_pool_fwd / _bwd: the
two-launch online-softmax forward for D % 8 == 0, the three-launch form otherwise) against float64: random numbers of ragged segments
(1 .. 20 000 rows, incl. 1-row segments), widths 64 .. 512, score ranges from flat to +-60 (so that the per-workgroup maxima differ by
far more than the exponent range of the weights), a row pitch larger than the width. usage: pool_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import ops  # noqa: E402

dev = "cuda:0"
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
worst = [0.0, 0.0, 0.0]
for case in range(ncase):
    D = rnd.choice((64, 128, 384, 384, 200, 512, 132, 68))
    nseg = rnd.randint(1, 17)
    lens = [rnd.choice((1, 2, 31, 32, 33, 1024)) if rnd.random() < 0.25 else rnd.randint(1, rnd.choice((300, 3000, 20000))) for _ in range(nseg)]
    N = sum(lens)
    span = rnd.choice((0.0, 1.0, 8.0, 60.0))
    s = torch.randn(N, generator=g) * span
    if span >= 8.0 and rnd.random() < 0.5:           # a trend along the rows: every workgroup sees another maximum
        s = s + torch.linspace(-span, span, N)
    pitch = D + rnd.choice((0, 0, 8, 64))
    hb = torch.randn(N, pitch, generator=g)
    h = hb[:, :D]
    seg = ops.Segments(lens, dev) if (nseg > 1 or rnd.random() < 0.5) else None
    hd = hb.to(dev)[:, :D]
    A, pooled = ops.softmax_pool(s.to(dev), hd, N, D, seg)
    dp = torch.randn(nseg, D, generator=g)
    dA = torch.randn(N, generator=g) if rnd.random() < 0.5 else None
    ds = ops.softmax_pool_bwd(dp.to(dev), None if dA is None else dA.to(dev), A, hd, N, D, seg)
    torch.cuda.synchronize()
    # float64 reference
    r0, eA, eP, eS = 0, 0.0, 0.0, 0.0
    for b, L in enumerate(lens):
        sb = s[r0:r0 + L].double().requires_grad_(True)
        a = torch.softmax(sb, dim=0)
        pb = a @ h[r0:r0 + L].double()
        loss = (pb * dp[b].double()).sum() + (0 if dA is None else (a * dA[r0:r0 + L].double()).sum())
        loss.backward()
        eA = max(eA, float((A[r0:r0 + L].cpu().double() - a.detach()).abs().max()))
        eP = max(eP, float((pooled[b].cpu().double() - pb.detach()).abs().max() / (pb.detach().abs().max() + 1e-30)))
        # ds_n = A_n (t_n - c), t = dA + h dp, c = sum A t: with peaked scores the difference cancels to ~0 -> measure against the
        # size of the terms before they cancel
        t = h[r0:r0 + L].double() @ dp[b].double() + (0 if dA is None else dA[r0:r0 + L].double())
        scale = float((a.detach() * (t.abs() + float((a.detach() * t).sum().abs()))).max()) + 1e-30
        eS = max(eS, float((ds[r0:r0 + L].cpu().double() - sb.grad).abs().max()) / scale)
        r0 += L
    worst = [max(worst[0], eA), max(worst[1], eP), max(worst[2], eS)]
    ok = eA < 2e-6 and eP < 5e-6 and eS < 2e-5 and bool(torch.isfinite(pooled).all()) and bool(torch.isfinite(ds).all())
    if not ok or case % 10 == 0:
        print(f"case {case}: D {D} pitch {pitch} segments {lens if nseg <= 6 else str(lens[:6]) + '...'} span {span}: A {eA:.1e} pooled {eP:.1e} ds {eS:.1e} "
              f"{'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print("all ok;", ncase, "cases; worst |A - ref|, pooled, ds relative:", worst)
