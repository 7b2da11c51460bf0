#!/usr/bin/env python3
"""Randomised check of the GENConv softmax-aggregation gather kernels (ops.genconv_aggregate -> advmil_genconv_fwd / _bwd over the two
CSR images of the graph) against float64 on RANDOM graphs -- not only the 8-NN grids of the WSI pipeline: random in-degrees 0 .. 40
(isolated nodes, hubs, self loops, multi-edges), channel widths 64 / 128 / 256, temperatures 0.3 .. 3, feature scales up to 2, forward
and both gradients (x, t). The aggregation restated in float64 is the published GENConv message passing (oracle header: PARITY
UNPINNED against torch_geometric itself). usage: graph_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import ops  # noqa: E402

dev = "cuda:0"
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))


def ref(x, t, ei, eps=1e-7):
    src, dst = ei[0], ei[1]
    n, d = x.shape
    msg = torch.relu(x[src]) + eps
    z = msg * t
    zmax = torch.full((n, d), -float("inf"), dtype=x.dtype).scatter_reduce(0, dst[:, None].expand(-1, d), z, reduce="amax", include_self=True)
    e = torch.exp(z - zmax[dst])
    den = torch.zeros(n, d, dtype=x.dtype).index_add_(0, dst, e)
    w = e / den[dst]
    return torch.zeros(n, d, dtype=x.dtype).index_add_(0, dst, w * msg) + x


worst = [0.0, 0.0, 0.0]
for case in range(ncase):
    N = rnd.choice((1, 7, 64, 500, 4096, rnd.randint(2, 9000)))
    C = rnd.choice((64, 128, 128, 256))
    kind = rnd.choice(("sparse", "dense", "hub", "none"))
    deg = {"sparse": 3, "dense": 24, "hub": 8, "none": 0}[kind]
    E = N * deg
    if E:
        src = torch.randint(0, N, (E,), generator=g)
        dst = torch.randint(0, N, (E,), generator=g)
        if kind == "hub":                             # a few targets collect a third of all edges
            dst[: E // 3] = torch.randint(0, max(1, N // 50), (E // 3,), generator=g)
        ei = torch.stack([src, dst])
    else:
        ei = torch.zeros(2, 0, dtype=torch.long)
    x = torch.randn(N, C, generator=g) * rnd.choice((0.5, 1.0, 2.0))       # (scale 4 x t 3 puts z = t msg at 40: the hubs' dx then reaches 1e-5)
    t = torch.tensor([rnd.uniform(0.3, 3.0)])
    go = torch.randn(N, C, generator=g)
    xd, td = x.clone().to(dev).requires_grad_(True), t.clone().to(dev).requires_grad_(True)
    csr = ops.GraphCSR(ei.to(dev), N)
    out = ops.genconv_aggregate(xd, td, csr)
    (out * go.to(dev)).sum().backward()
    xr, tr = x.clone().double().requires_grad_(True), t.clone().double().requires_grad_(True)
    orf = ref(xr, tr, ei)
    (orf * go.double()).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / (b.abs().max() + 1e-30))
    e_o, e_x = rel(out, orf.detach()), rel(xd.grad, xr.grad)
    # dt = sum over (edge, channel) of dout w msg (msg - sum w msg): signed terms that largely cancel -> measure the error against the
    # sum of their magnitudes (what an fp32 accumulation can resolve), not against the cancelled total
    with torch.no_grad():
        if ei.shape[1]:
            src_, dst_ = ei[0], ei[1]
            msg = torch.relu(x.double()[src_]) + 1e-7
            z = msg * t.double()
            zmax = torch.full((N, C), -float("inf"), dtype=torch.float64).scatter_reduce(0, dst_[:, None].expand(-1, C), z, reduce="amax", include_self=True)
            e_ = torch.exp(z - zmax[dst_])
            w_ = e_ / torch.zeros(N, C, dtype=torch.float64).index_add_(0, dst_, e_)[dst_]
            agg = torch.zeros(N, C, dtype=torch.float64).index_add_(0, dst_, w_ * msg)
            # (the uncancelled products w msg^2 and w msg agg: the kernel forms the difference of two fp32 sums)
            mag = float((go.double()[dst_].abs() * w_ * msg * (msg + agg[dst_])).sum())
        else:
            mag = 0.0
    e_t = abs(float(td.grad) - float(tr.grad)) / (mag + 1e-30) if mag > 0 else abs(float(td.grad) - float(tr.grad))
    worst = [max(worst[0], e_o), max(worst[1], e_x), max(worst[2], e_t)]
    ok = e_o < 4e-6 and e_x < 2e-5 and e_t < 2e-6       # (fp32 sums over 100-400 in-edges at the hubs: dx 1e-5 there, 4e-7 on grids)
    if not ok or case % 10 == 0:
        print(f"case {case}: {kind} graph N {N} E {E} C {C} t {float(t):.2f}: out {e_o:.1e} dx {e_x:.1e} dt {e_t:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print("all ok;", ncase, "cases; worst out / dx / dt:", worst)
