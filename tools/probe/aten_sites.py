#!/usr/bin/env python3
"""Python call sites of the torch operators one eager optimizer step still issues (cat / copy_ / fill / add / mul / gather ...): a
TorchDispatchMode logs every ATen call that launches a kernel together with the innermost advmil_amd frame.
usage: aten_sites.py [mode=abmil] [patches=8192] [bags=16]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "abmil"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bags = int(sys.argv[3]) if len(sys.argv) > 3 else 16
case = bench.Case(torch, torch.device("cuda", 0), kind, n, bags, max(16, bags), "bf16x3", seed=1, eager=True)
for _ in range(2):
    case.eager_step()
torch.cuda.synchronize()
SKIP = ("view", "reshape", "slice", "select", "t.", "transpose", "expand", "as_strided", "detach", "alias", "empty", "unsqueeze", "squeeze",
        "permute", "narrow", "split", "unbind", "_unsafe_view", "contiguous", "_to_copy", "lift_fresh", "item", "_local_scalar_dense",
        "resize_", "set_", "chunk", "flatten", "stride", "size", "numel", "sym_", "record_stream", "is_pinned", "_has_compatible")
seen = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if "advmil_amd" in f.filename]
            where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            seen[(name, str(shp)[:50], where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    case.eager_step()
torch.cuda.synchronize()
print("ATen calls of one step (non-view):", sum(seen.values()))
for (name, shp, where), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d}x {name:34s} {shp:52s} {where}")
