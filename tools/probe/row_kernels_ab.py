#!/usr/bin/env python3
"""Same-box timing of the gated-attention row kernels at the 16-bag slab (131072 x 384): gate_score, gate_bwd, dropout of planes, pooling
backward from planes. Run once per library build (ADVMIL_HIP_LIB=... for the other one) in the SAME gpurun call:
  python tools/probe/row_kernels_ab.py; ADVMIL_HIP_LIB=build_alt/lib_old.so python tools/probe/row_kernels_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from advmil_amd import ops

dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
N, D, bags = 131072, 384, 16
nb = 3
abs_ = [torch.rand(N, 2 * D, device=dev) for _ in range(nb)]
hs = [ops.split_planes(torch.randn(N, D, device=dev).relu_()) for _ in range(nb)]
wc, bc = torch.randn(D, device=dev) * 0.05, torch.zeros(1, device=dev)
ds = torch.randn(N, device=dev)
rng = ops.DeviceRng(dev, seed=3)
seg = ops.Segments([N // bags] * bags, dev)
A = torch.rand(N, device=dev)
dp = torch.randn(bags, D, device=dev)
dA = torch.randn(N, device=dev)
tok = torch.empty(N, D, device=dev)


def t(tag, fn, iters=30):
    for k in range(3):
        fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:28s} {e0.elapsed_time(e1) * 1e3 / iters:7.1f} us", flush=True)


print("library:", os.environ.get("ADVMIL_HIP_LIB", "in-tree"))
with torch.no_grad():
    t("gate_score (p = 0.25)", lambda k: ops.gate_score(abs_[k % nb], wc, bc, N, D, 0.25, rng.seed, 3, 4, None))
    t("gate_score (p = 0)", lambda k: ops.gate_score(abs_[k % nb], wc, bc, N, D))
    gpl = ops.Planes.alloc((N, 2 * D), dev)
    t("gate_bwd (planes only)", lambda k: ops.gate_bwd(abs_[k % nb], ds, wc, N, D, 0.25, rng.seed, 3, 4, planes=gpl, planes_only=True))
    t("dropout_planes", lambda k: ops.dropout_planes(hs[k % nb], N, D, 0.25, rng.seed, 5, None))
    t("pool_bwd from planes", lambda k: ops.softmax_pool_bwd(dp, dA, A, tok, N, D, seg, hs[k % nb]))
    s1 = ops.gate_score(abs_[0], wc, bc, N, D, 0.25, rng.seed, 3, 4, None)
    print("score checksum", float(s1.double().sum()), float(s1.double().abs().max()))
