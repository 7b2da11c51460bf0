#!/usr/bin/env python3
"""Which torch (ATen) operators one eager optimizer step still issues, with shapes and Python call sites (CPU-side profile)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
dev = torch.device("cuda", 0)
case = bench.Case(torch, dev, "abmil", 8192, 16, 16, "bf16x3", seed=1, eager=True)
for _ in range(2):
    case.eager_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    case.eager_step()
    torch.cuda.synchronize()
VIEWS = {"aten::" + n for n in ("view", "reshape", "slice", "select", "t", "transpose", "expand", "as_strided", "detach", "alias", "empty",
                                  "empty_like", "empty_strided", "unsqueeze", "squeeze", "permute", "narrow", "split", "split_with_sizes",
                                  "unbind", "_unsafe_view", "view_as", "contiguous", "to", "_to_copy", "lift_fresh", "detach_", "item",
                                  "_local_scalar_dense", "result_type", "is_nonzero", "resize_", "set_", "chunk", "flatten", "unflatten",
                                  "expand_as", "numpy_T", "real", "conj", "resolve_conj", "resolve_neg", "stride", "size", "numel")}
evs = [e for e in prof.events() if e.name.startswith("aten::") and e.name not in VIEWS and not [c for c in e.cpu_children if c.name.startswith("aten::") and c.name not in VIEWS]]
agg = {}
for e in evs:
    st = [f for f in (e.stack or []) if "advmil_amd" in f or "bench.py" in f]
    key = (e.name, str(e.input_shapes)[:60], (st[0].split("advmil_amd/")[-1] if st else "?")[:70])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e.cpu_time
tot = sum(v[0] for v in agg.values())
print("leaf aten ops (kernel-launching kinds):", tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{v[0]:3d}x {v[1]:7.1f} us  {k[0]:28s} {k[1]:62s} {k[2]}")
