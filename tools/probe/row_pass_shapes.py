#!/usr/bin/env python3
"""Which tensors the row passes of one eager optimizer step touch: every call of ops.act_dropout_bwd / gate_bwd / gate_score / ln_relu
wrappers with its shape and role (forward dropout replay over a memoized activation vs activation backward), in call order, with the
caller's frame. usage: row_pass_shapes.py [mode=abmil] [patches=8192] [bags=16]"""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from advmil_amd import ops  # noqa: E402
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "abmil"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bags = int(sys.argv[3]) if len(sys.argv) > 3 else 16
case = bench.Case(torch, dev, kind, n, bags, max(16, bags), "bf16x3", seed=1, eager=True)
for _ in range(2):
    case.eager_step()
torch.cuda.synchronize()
log = []


def where():
    out = []
    for fr in traceback.extract_stack()[:-3]:
        if "advmil_amd" in fr.filename and fr.name not in ("apply",):
            out.append(f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}")
    return " < ".join(out[-4:][::-1])


def wrap(name):
    f = getattr(ops, name)

    def g(*a, **k):
        shp = [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]
        ints = [t for t in a if isinstance(t, (int, float))][:4]
        log.append(f"{name:18s} {shp} {ints} {({kk: (vv if not torch.is_tensor(vv) and not hasattr(vv, 'hi') else 'T') for kk, vv in k.items()})}  @ {where()}")
        return f(*a, **k)
    setattr(ops, name, g)


for nm in ("act_dropout_bwd", "gate_bwd", "gate_score", "colsum"):
    wrap(nm)
case.eager_step()
torch.cuda.synchronize()
print("\n".join(log))
