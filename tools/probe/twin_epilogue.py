#!/usr/bin/env python3
"""Cost of the two-layer launch's epilogue variants at the headline shape (131072 x (384 + 128) x 1024), back to back with a memory-bound
pass in between (the launch never runs back to back in the step): fp32 + planes (round 5) | planes only | planes + train-mode twin + bits.
Also checks the twin against the replay kernel it replaces (bitwise: same draw, same split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from advmil_amd import ops

dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
M, K, N1, N2 = 131072, 1024, 384, 128
xs = [torch.randn(M, K, device=dev) for _ in range(2)]
xpl = [ops.split_planes(x) for x in xs]
W1, W2 = torch.randn(N1, K, device=dev) * 0.03, torch.randn(N2, K, device=dev) * 0.03
b1, b2 = torch.randn(N1, device=dev) * 0.1, torch.randn(N2, device=dev) * 0.1
w1pl, w2pl = ops.split_planes(W1), ops.split_planes(W2)
rng = ops.DeviceRng(dev, seed=11)
cool = torch.empty(1 << 26, device=dev)


def t(tag, fn, iters=10):
    st = ops.Stamps(dev)
    for _ in range(2):
        fn(0)
    torch.cuda.synchronize()
    ops.STAMPS = st
    for k in range(iters):
        cool.mul_(1.0)
        fn(k)
    ops.STAMPS = None
    torch.cuda.synchronize()
    us = [u for _, _, _, u in st.durations_us()]
    print(f"{tag:44s} {sum(us) / len(us):7.1f} us  (min {min(us):.1f}, max {max(us):.1f})", flush=True)


with torch.no_grad():
    t("fp32 + planes (round 5)", lambda k: ops.gemm_two_layers(xs[k % 2], xpl[k % 2], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, emit_planes1=True))
    t("planes only", lambda k: ops.gemm_two_layers(xs[k % 2], xpl[k % 2], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, y1_planes_only=True))
    t("planes only + twin + bits", lambda k: ops.gemm_two_layers(xs[k % 2], xpl[k % 2], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, y1_planes_only=True,
                                                                 twin=(0.25, rng.seed, 7, None)))
    t("fp32 + planes + twin + bits", lambda k: ops.gemm_two_layers(xs[k % 2], xpl[k % 2], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, emit_planes1=True,
                                                                   twin=(0.25, rng.seed, 7, None)))
    # parity of the twin with the replay kernel
    y1, y2, cpl = ops.gemm_two_layers(xs[0], xpl[0], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, emit_planes1=True, twin=(0.25, rng.seed, 7, None))
    tpl, tbits = cpl.twin
    rpl = ops.Planes.alloc((M, N1), dev)
    rbits = torch.empty(M, N1 // 32, dtype=torch.int32, device=dev)
    yr, _ = ops.act_dropout_bwd(y1, y1, ops.ACT_NONE, M, N1, 0.25, rng.seed, 7, want_bias=False, planes=rpl, bits=rbits)
    torch.cuda.synchronize()
    print("twin == replay: hi", bool(torch.equal(tpl.hi, rpl.hi)), "lo", bool(torch.equal(tpl.lo, rpl.lo)), "bits", bool(torch.equal(tbits, rbits)),
          "kept fraction", float((tpl.hi != 0).float().mean()))
    y1b, y2b, cplb = ops.gemm_two_layers(xs[0], xpl[0], W1, w1pl, b1, ops.ACT_RELU, W2, w2pl, b2, ops.ACT_NONE, y1_planes_only=True)
    torch.cuda.synchronize()
    print("planes-only == planes beside fp32:", bool(torch.equal(cplb.hi, cpl.hi) and torch.equal(cplb.lo, cpl.lo)), "y2 equal", bool(torch.equal(y2, y2b)))
