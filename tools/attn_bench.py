#!/usr/bin/env python3
"""Time the fused ESAT attention core (advmil_mha_fwd / advmil_mha_bwd) on a slab of equal-length bags.
usage: attn_bench.py [L=2048] [bags=16] [p=0.25] [iters=20]
Algorithmic flops: forward 4*L^2*d per bag (QK^T + PV, d = 384), backward 10*L^2*d (five contractions)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = "cuda:0"
d, nh = 384, 8
qkv = torch.randn(G * L, 3 * d, device=dev, requires_grad=True)
go = torch.randn(G * L, d, device=dev)
seg = ops.Segments([L] * G, dev)
rng = ops.DeviceRng(dev, seed=1)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


with torch.no_grad():
    us_f = timed(lambda: ops.mha(qkv, nh, p, rng, seg=seg), iters)
o = ops.mha(qkv, nh, p, rng, seg=seg)


def bwd():
    qkv.grad = None
    o.backward(go, retain_graph=True)


us_b = timed(bwd, iters)
ff, fb = 4.0 * L * L * d * G, 10.0 * L * L * d * G
print(f"L={L} bags={G} p={p}: fwd {us_f:.1f} us = {ff / us_f / 1e6:.1f} TF ({ff / us_f / 1e6 / 833.3:.3f} of bf16x3 roof); "
      f"bwd (prep + dQ + dKdV) {us_b:.1f} us = {fb / us_b / 1e6:.1f} TF ({fb / us_b / 1e6 / 833.3:.3f})")
