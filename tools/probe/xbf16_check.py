#!/usr/bin/env python3
"""Dev check: contractions over a bf16 slab (x_storage = 'bf16': the slab is ONE bf16 plane, two MFMAs per product) against float64 on
the same bf16-rounded values, on every kernel the mode can land on; then times the three slab contractions of the ABMIL step in both
storage modes. usage: xbf16_check.py [rows=131072]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops  # noqa: E402

ops.set_gemm_mode("bf16x3")
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
g = torch.Generator(device=dev).manual_seed(3)
K = 1024
x = torch.randn(M, K, device=dev, generator=g)
xb = x.to(torch.bfloat16)
xd = xb.double()


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


def timed(fn, n=20):
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


xpl_full = ops.split_planes(x)
for N in (128, 384, 512):
    W = torch.randn(N, K, device=dev, generator=g) / 32
    wpl = ops.split_planes(W)
    ref = (xd[:8192] @ W.double().t())
    y = ops.gemm(xb, W, True, True, M, N, K, b_planes=wpl)
    y3 = ops.gemm(x, W, True, True, M, N, K, a_planes=xpl_full, b_planes=wpl)
    t2 = timed(lambda: ops.gemm(xb, W, True, True, M, N, K, b_planes=wpl))
    t3 = timed(lambda: ops.gemm(x, W, True, True, M, N, K, a_planes=xpl_full, b_planes=wpl))
    print(f"NT {M}x{N}x{K}: bf16 slab vs float64 {rel(y[:8192], ref):.2e}   {t2:.1f} us (bf16 slab) vs {t3:.1f} us (hi+lo slab)  "
          f"[fp32-slab result vs bf16-slab result: {rel(y3[:8192], y[:8192].double()):.2e}]", flush=True)
# two layers in one launch
W1 = torch.randn(384, K, device=dev, generator=g) / 32
W2 = torch.randn(128, K, device=dev, generator=g) / 32
b1, b2 = torch.randn(384, device=dev, generator=g), torch.randn(128, device=dev, generator=g)
p1, p2 = ops.split_planes(W1), ops.split_planes(W2)
if ops.gemm_two_layers_ok(M, 384, 128, K):
    y1, y2, _ = ops.gemm_two_layers(xb, ops.Planes(xb, None), W1, p1, b1, 1, W2, p2, b2, 0)
    r1 = torch.relu(xd[:8192] @ W1.double().t() + b1.double())
    r2 = xd[:8192] @ W2.double().t() + b2.double()
    t2 = timed(lambda: ops.gemm_two_layers(xb, ops.Planes(xb, None), W1, p1, b1, 1, W2, p2, b2, 0))
    t3 = timed(lambda: ops.gemm_two_layers(x, xpl_full, W1, p1, b1, 1, W2, p2, b2, 0))
    print(f"two layers {M}x(384+128)x{K}: {rel(y1[:8192], r1):.2e} / {rel(y2[:8192], r2):.2e}   {t2:.1f} us (bf16 slab) vs {t3:.1f} us", flush=True)
# weight gradients dY^T X
for N in (384, 128):
    dy = torch.randn(M, N, device=dev, generator=g)
    dpl = ops.split_planes(dy)
    ref = dy.double().t() @ xd
    dW = ops.gemm(None, xb, False, False, N, K, M, a_planes=dpl)
    dWg = ops.gemm(dy, xb, False, False, N, K, M)
    t2 = timed(lambda: ops.gemm(None, xb, False, False, N, K, M, a_planes=dpl))
    t3 = timed(lambda: ops.gemm(None, x, False, False, N, K, M, a_planes=dpl, b_planes=xpl_full))
    t2g = timed(lambda: ops.gemm(dy, xb, False, False, N, K, M))
    t3g = timed(lambda: ops.gemm(dy, x, False, False, N, K, M, b_planes=xpl_full))
    print(f"TN {N}x{K}x{M}: planes/planes {rel(dW, ref):.2e} ({t2:.1f} vs {t3:.1f} us)   fp32 dY / bf16 slab {rel(dWg, ref):.2e} ({t2g:.1f} vs {t3g:.1f} us)",
          flush=True)
# a small slab through linear_act (generic kernels, autograd)
xs = xb[:512].contiguous()
W = (torch.randn(384, K, device=dev, generator=g) / 32).requires_grad_(True)
b = torch.randn(384, device=dev, generator=g).requires_grad_(True)
y = ops.linear_act(xs, W, b, "relu")
go = torch.randn_like(y)
y.backward(go)
Wd, bd = W.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
yr = torch.relu(xs.double() @ Wd.t() + bd)
yr.backward(go.double())
print(f"small slab linear_act: y {rel(y, yr):.2e}  dW {rel(W.grad, Wd.grad):.2e}  db {rel(b.grad, bd.grad):.2e}")
