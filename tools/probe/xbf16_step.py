#!/usr/bin/env python3
"""Dev check: the ABMIL / ESAT step with bags held as one bf16 plane (x_storage = 'bf16') against fp32 storage: bags/s of both on the
same box, and the deviation of the step's outputs (bench.xbf16_parity). usage: xbf16_step.py [mode=abmil] [patches=8192] [steps=40]"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "abmil"
patches = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = types.SimpleNamespace(mode=mode, patches=patches, bags=16, pool=64 if patches <= 8192 else 16, gemm_mode="bf16x3")


def sync():
    torch.cuda.synchronize()


for xs in ("fp32", "bf16", "fp32", "bf16"):
    c = bench.Case(torch, dev, mode, patches, args.bags, args.pool, "bf16x3", 1234, x_storage=xs)
    dt, _ = c.timed(steps, 5, sync)
    print(f"x_storage={xs}: {args.bags * steps / dt:.1f} bags/s  {1e3 * dt / steps:.3f} ms/step  finite={c.logs_finite()}  {c.launch_note}", flush=True)
    c.free()
    del c
print(bench.xbf16_parity(torch, dev, args))
