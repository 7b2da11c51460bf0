#!/usr/bin/env python3
"""Time the LayerNorm-ReLU-mean16 region-embedding tail (fwd / bwd) on a slab: usage ln_bench.py [rows=524288] [d=384]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = "cuda:0"
y = torch.randn(rows, d, device=dev); g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
emb, mean, rstd = ops.ln_relu_mean16_fwd(y, g, b, rows, d)
demb = torch.randn_like(emb)
uf = t(lambda: ops.ln_relu_mean16_fwd(y, g, b, rows, d))
ub = t(lambda: ops.ln_relu_mean16_bwd(demb, y, g, b, mean, rstd, rows, d))
by = rows * d * 4
print(f"rows {rows} d {d}: fwd {uf:.0f} us = {by / uf / 1e3:.0f} GB/s ({by / uf / 8e6:.2f} of 8 TB/s); bwd {ub:.0f} us = {2 * by / ub / 1e3:.0f} GB/s ({2 * by / ub / 8e6:.2f})")
