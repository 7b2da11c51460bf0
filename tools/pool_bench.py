#!/usr/bin/env python3
"""Launch the attention-pool forward and backward calls (advmil_softmax_pool_fwd / _bwd) N times on rotating slabs, for
`rocprofv3 --kernel-trace --stats` (per-kernel durations) and the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes.
usage: pool_bench.py [patches=8192] [bags=16] [iters=40] [planes=1]
planes = 1 (the product's form since round 6): h is read as its bf16x3 operand planes (hi + lo: the same 4 bytes per element); 0: fp32 rows.
Algorithmic bytes per pool_partial8 launch: rows x 384 x 4 (h read once) + rows x 4 (scores)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd import ops  # noqa: E402

patches = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bags = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
planes = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
dev, D = "cuda:0", 384
rows = patches * bags
nbuf = max(2, int(600e6 // (4 * rows * D)) + 1)       # rotate slabs past the 256 MB Infinity Cache
hs = [torch.randn(rows, D, device=dev) for _ in range(nbuf)]
pls = [ops.split_planes(h) for h in hs] if planes else [None] * nbuf
s = torch.randn(rows, device=dev)
seg = ops.Segments([patches] * bags, dev)
dp = torch.randn(bags, D, device=dev)
for k in range(iters):
    A, pooled = ops.softmax_pool(s, hs[k % nbuf], rows, D, seg, pls[k % nbuf])
    ops.softmax_pool_bwd(dp, None, A, hs[(k + 1) % nbuf], rows, D, seg, pls[(k + 1) % nbuf])
torch.cuda.synchronize()
print("done rows", rows, "bags", bags, "slabs", nbuf, "planes", planes, "bytes_per_pass", 4 * rows * D + 4 * rows)
