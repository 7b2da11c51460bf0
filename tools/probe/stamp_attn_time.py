#!/usr/bin/env python3
"""Read the forward attention kernel's section stamps (tools/probe/stamp_attn.sh run): per tile trip of waves 0 and 4 of workgroup 0,
cycles spent in [DMA issue | half-step 0 | half-step 1 | vmcnt wait | barrier]. usage: stamp_attn_time.py [L=2048] [bags=16] [p=0.25]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import _lib, ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
dev = "cuda:0"
qkv = torch.randn(G * L, 3 * 384, device=dev)
seg = ops.Segments([L] * G, dev)
rng = ops.DeviceRng(dev, seed=1)
with torch.no_grad():
    for _ in range(3):
        ops.mha(qkv, 8, p, rng, seg=seg)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(h, "advmil_debug_attn_occupancy"):
    print("occupancy API (workgroups per CU): fwd", h.advmil_debug_attn_occupancy(0), "dq", h.advmil_debug_attn_occupancy(1), "dkv", h.advmil_debug_attn_occupancy(2))
buf = (ctypes.c_ulonglong * 2048)()
assert h.advmil_debug_attn_stamps(buf, 2048) == 0
for w in range(2):
    st = [(int(v) >> 4, int(v) & 15) for v in buf[w * 1024:(w + 1) * 1024] if v]
    print(f"wave {4 * w}: {len(st)} stamps")
    names = {2: "issue", 3: "half0", 4: "half1", 5: "vmwait", 1: "barrier", 6: "barrier"}
    tot = {}
    rows = []
    for (t0, _), (t1, g1) in zip(st[:-1], st[1:]):
        tot.setdefault(names[g1], []).append(t1 - t0)
    for k, v in tot.items():
        v2 = v[2:-2] if len(v) > 6 else v
        print(f"   {k:8s} mean {sum(v2) / len(v2):8.0f} cycles  (min {min(v2)}, max {max(v2)}, n {len(v2)})")
    per = [sum(x) for x in zip(*[tot[k][:min(len(u) for u in tot.values())] for k in tot])]
    print(f"   trip total mean {sum(per[2:-2]) / max(len(per[2:-2]), 1):.0f} cycles")
