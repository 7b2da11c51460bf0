#!/usr/bin/env python3
"""dh = dG Wab (NN) and dWab = dG^T h (TN) with dG taken from planes (A operand pre-split) vs split on the fly."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev="cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/iters
R, D = 131072, 384
dG = torch.randn(R, 2*D, device=dev); Wab = torch.randn(2*D, D, device=dev)*0.05; h = torch.randn(R, D, device=dev)
pg = ops.split_planes(dG)
o0 = ops.gemm(dG, Wab, True, False, R, D, 2*D); o1 = ops.gemm(dG, Wab, True, False, R, D, 2*D, a_planes=pg)
print("dh plan", ops.gemm_plan(R, D, 2*D, True, False), "bit-identical", torch.equal(o0, o1),
      f"fly {bench(lambda: ops.gemm(dG, Wab, True, False, R, D, 2*D)):.0f} us  planes {bench(lambda: ops.gemm(dG, Wab, True, False, R, D, 2*D, a_planes=pg)):.0f} us")
w0 = ops.gemm(dG, h, False, False, 2*D, D, R); w1 = ops.gemm(dG, h, False, False, 2*D, D, R, a_planes=pg)
print("dWab plan", ops.gemm_plan(2*D, D, R, False, False), "bit-identical", torch.equal(w0, w1),
      f"fly {bench(lambda: ops.gemm(dG, h, False, False, 2*D, D, R)):.0f} us  planes {bench(lambda: ops.gemm(dG, h, False, False, 2*D, D, R, a_planes=pg)):.0f} us")
