#!/usr/bin/env python3
"""Why does the pooling forward take ~50 us inside the step and ~33 us standalone? Device-clock stamps around the pooling
call (a) alone, rotating slabs; (b) right behind the fused gate-score contraction, as in the step; (c) behind the contraction with an
idle gap of G us in between (does the clock come back?); (d) behind a memory-bound neighbour instead of the MFMA one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from advmil_amd import ops

dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
N, D, bags = 131072, 384, 16
nbuf = 3
hs = [torch.randn(N, D, device=dev).relu_() for _ in range(nbuf)]
hpl = [ops.split_planes(h) for h in hs]
for h, p in zip(hs, hpl):
    h._advmil_planes = p
Wa, Wb = torch.randn(D, D, device=dev) * 0.05, torch.randn(D, D, device=dev) * 0.05
ba, bb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
wc, bc = torch.randn(1, D, device=dev) * 0.05, torch.zeros(1, device=dev)
seg = ops.Segments([N // bags] * bags, dev)
sc = torch.randn(N, device=dev)
big = torch.empty(1 << 27, device=dev)     # 512 MB: a memory-bound neighbour


def run(tag, fn, iters=12):
    st = ops.Stamps(dev)
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    ops.STAMPS = st
    try:
        for k in range(iters):
            fn(k)
    finally:
        ops.STAMPS = None
    torch.cuda.synchronize()
    acc = {}
    for name, shape, fl, us in st.durations_us():
        acc.setdefault(name, []).append(us)
    print(tag, {k: (round(sum(v) / len(v), 1), round(min(v), 1), round(max(v), 1), len(v)) for k, v in acc.items()}, flush=True)


with torch.no_grad():
    run("(a) pool alone, rotating slabs           ", lambda k: ops.softmax_pool(sc, hs[k % nbuf], N, D, seg))
    run("(b) gate contraction -> pool (as in step)", lambda k: ops.gated_attn_pool(hs[k % nbuf], Wa, ba, Wb, bb, wc, bc, seg=seg))
    for gap_us in (20, 100, 400):
        def f(k, gap_us=gap_us):
            h = hs[k % nbuf]
            Wi, bi, wipl = ops.gate_interleave(Wa, ba, Wb, bb, D, planes=True)
            s = ops.gate_partial_sum(ops.gemm(h, Wi, True, True, N, 2 * D, D, bias=bi, gate_wc=wc.reshape(-1), a_planes=h._advmil_planes, b_planes=wipl), bc)
            torch.cuda._sleep(int(gap_us * 2100))
            ops.softmax_pool(s, h, N, D, seg)
        run(f"(c) contraction, idle {gap_us:4d} us, pool       ", f)

    def g(k):
        big.mul_(1.0001)
        ops.softmax_pool(sc, hs[k % nbuf], N, D, seg)
    run("(d) 1 GB elementwise pass -> pool        ", g)
    # pool twice in a row behind the contraction: is the second call faster?
    def h2(k):
        h = hs[k % nbuf]
        ops.gated_attn_pool(h, Wa, ba, Wb, bb, wc, bc, seg=seg)
        ops.softmax_pool(sc, hs[(k + 1) % nbuf], N, D, seg)
        ops.softmax_pool(sc, hs[(k + 2) % nbuf], N, D, seg)
    run("(e) contraction -> pool -> pool -> pool  ", h2)
    # (f) as in the step: a kernel WRITES ~470 MB (the two-layer launch's outputs), the contraction reads, then the pool
    outs = [torch.empty(N, D, device=dev) for _ in range(2)] + [torch.empty(N, 128, device=dev)]
    def f2(k):
        h = hs[k % nbuf]
        for o in outs:
            o.fill_(1.0)                         # 201 + 201 + 67 MB of fresh dirty lines
        ops.gated_attn_pool(h, Wa, ba, Wb, bb, wc, bc, seg=seg)
    run("(f) 470 MB written, contraction, pool    ", f2)
    def f3(k):
        h = hs[k % nbuf]
        h.mul_(1.0)                              # h itself freshly written (dirty in L2 / Infinity Cache), as the two-layer launch leaves it
        outs[1].fill_(1.0); outs[2].fill_(1.0)
        ops.gated_attn_pool(h, Wa, ba, Wb, bb, wc, bc, seg=seg)
    run("(g) h rewritten + 268 MB, contraction, pool", f3)
