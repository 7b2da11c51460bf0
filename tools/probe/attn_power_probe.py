#!/usr/bin/env python3
"""Dev probe: is the fused attention core bound by its instruction stream or by the chip's power budget?
Runs advmil_mha_fwd / bwd at 16 x 2048 tokens with (a) random operands, (b) zero operands (same instruction stream, no operand
toggling), (c) dropout on / off, and prints PER-LAUNCH times of a back-to-back burst that follows an idle pause (a cold chip clocks
higher than one that has run the same launch for a second).
usage: attn_power_probe.py [L=2048] [bags=16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda:0"
d, nh = 384, 8
seg = ops.Segments([L] * G, dev)
rng = ops.DeviceRng(dev, seed=1)


def burst(fn, n, pause):
    fn()
    torch.cuda.synchronize()
    time.sleep(pause)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]


def fmt(ts):
    return " ".join(f"{t:.0f}" for t in ts)


for name, scale in (("randn", 1.0), ("zeros", 0.0), ("randn*0.05", 0.05)):
    qkv = (torch.randn(G * L, 3 * d, device=dev) * scale).requires_grad_(True)
    go = torch.randn(G * L, d, device=dev) * (1.0 if scale else 0.0)
    for p in (0.25, 0.0):
        with torch.no_grad():
            ts = burst(lambda: ops.mha(qkv, nh, p, rng, seg=seg), 40, 1.0)
        print(f"{name:11s} p={p:4.2f} fwd(+split) first8: {fmt(ts[:8])} | last8: {fmt(ts[-8:])} | mean last20 {sum(ts[-20:]) / 20:.1f} us", flush=True)
        o = ops.mha(qkv, nh, p, rng, seg=seg)

        def bwd():
            qkv.grad = None
            o.backward(go, retain_graph=True)

        ts = burst(bwd, 40, 1.0)
        print(f"{name:11s} p={p:4.2f} bwd         first8: {fmt(ts[:8])} | last8: {fmt(ts[-8:])} | mean last20 {sum(ts[-20:]) / 20:.1f} us", flush=True)
