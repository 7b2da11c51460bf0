#!/usr/bin/env python3
"""In-graph time of the discriminator's bag-level tail, forward + backward, fused (advmil_dtail_*) against layer by layer.
usage: tail_time.py [B ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402
from advmil_amd.optim import FlatAdam  # noqa: E402
from tests.test_parity_gpu import DEV, build_disc, load_synth  # noqa: E402

ops.set_gemm_mode("bf16x3")
Bs = [int(v) for v in sys.argv[1:]] or [2, 4, 16, 32]


def graph_us(fn, reps=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 10)


for B in Bs:
    row = []
    for fused in (False, True):
        for frozen in (False, True):
            ops.DTAIL = fused
            d = build_disc("prj", "instance", "x").train()
            load_synth(d, "D-prj:")
            opt = FlatAdam(d, lr=1e-4)
            rng = ops.DeviceRng(DEV, seed=1)
            for m in d.modules():
                m.rng = rng
            eb = torch.randn(B, 128, device=DEV, requires_grad=not frozen)
            im = torch.randn(B, 128, device=DEV, requires_grad=not frozen)
            t = torch.rand(B, 1, device=DEV, requires_grad=True)
            if frozen:
                for p in d.parameters():
                    p.requires_grad_(False)
            one = torch.ones(B, 1, device=DEV)

            def step():
                f = d.tail(eb, im, t)
                torch.autograd.backward(f, grad_tensors=one)
            row.append(graph_us(step))
    print(f"B={B:3d}  layer-by-layer: D-phase {row[0]:6.1f} us  G-phase (frozen) {row[1]:6.1f} us   fused: D-phase {row[2]:6.1f} us  G-phase {row[3]:6.1f} us")
