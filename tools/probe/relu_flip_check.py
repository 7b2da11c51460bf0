#!/usr/bin/env python3
"""Is a raw-gradient deviation of a fuzz case a ReLU-boundary flip? For the configuration of tools/probe/oracle_fuzz.py 40 23 case 6
(ESAT, one bag of 400 patches, bce, bag-level inner product, no projection): the ORACLE alone, in fp32 and in float64, on the CPU -- no
HIP code involved. If the oracle's own fp32 and float64 gradients of `backbone.patch_embedding_layer.conv.weight` differ by what the HIP
path differs from float64 (1.9e-4 at a tensor scale of 1.26e-2), the deviation is a property of the input -- a LayerNorm output within
round-off of 0 takes the other ReLU branch -- not of a kernel. Also prints the smallest |LayerNorm output| of the region embedding.
With a GPU it also forms the same pre-activation with the HIP contraction (exact fp32 mode) and reports on which side of 0 the entry
lands there. usage: relu_flip_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import synth  # noqa: E402
from oracle import advmil_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

kind, n = "patch", 400
cfg = O.StepConfig(kind=kind, loss_netD="bce", l1_coef=0.0, disc_type="prj", inner_product="bag", prj_path=None)
PG = {k: H.T(synth.param(H.PARAM_SEED, f"G-{kind}:" + k, tuple(s))) for k, s in H.shapes_generator(kind).items()}
PD = {k: H.T(synth.param(H.PARAM_SEED, "D-prj:" + k, tuple(s))) for k, s in H.shapes_disc("prj", None).items()}
y = H.label(0); y[0, 1] = 1.0
bag = (H.bag(40, 512)[:, :n].contiguous(), None, y)
ng = [[H.noise_tensor("gr_g", 0, 192)]]
grads = {}
for name, cast in (("fp32", lambda t: t.float()), ("fp64", lambda t: t.double())):
    pg, pd = {k: cast(v) for k, v in PG.items()}, {k: cast(v) for k, v in PD.items()}
    b = [(cast(bag[0]), None, cast(bag[2]))]
    _, g, _ = O.update_gen(cfg, pg, pd, b, [[cast(ng[0][0])]])
    grads[name] = {k: v.double() for k, v in g.items()}
    W, bb = pg["backbone.patch_embedding_layer.conv.weight"], pg["backbone.patch_embedding_layer.conv.bias"]
    z = b[0][0].reshape(-1, 1024) @ W.reshape(W.shape[0], -1).t() + bb
    ln = F.layer_norm(z, (z.shape[1],), pg["backbone.patch_embedding_layer.norm.weight"], pg["backbone.patch_embedding_layer.norm.bias"], 1e-5)
    a = ln.abs()
    print(f"{name}: smallest |LayerNorm output| of the region embedding {float(a.min()):.3e}; entries below 1e-6: {int((a < 1e-6).sum())}, below 1e-5: {int((a < 1e-5).sum())} of {a.numel()}")
    grads[name + "_ln"] = ln.double()
k = "backbone.patch_embedding_layer.conv.weight"
d = (grads["fp32"][k] - grads["fp64"][k]).abs()
print(f"oracle fp32 vs oracle fp64, d/d {k}: max abs {float(d.max()):.3e} at a tensor scale of {float(grads['fp64'][k].abs().max()):.3e}")
flip = ((grads["fp32_ln"] > 0) != (grads["fp64_ln"] > 0))
print("LayerNorm outputs on different sides of 0 in fp32 and float64:", int(flip.sum()), "at", flip.nonzero()[:4].tolist(),
      "values", [f"{float(grads['fp32_ln'][tuple(i)]):.2e} / {float(grads['fp64_ln'][tuple(i)]):.2e}" for i in flip.nonzero()[:4]])

if torch.cuda.is_available():
    from advmil_amd import ops
    ops.set_gemm_mode("exact")
    dev = torch.device("cuda", 0)
    W = PG["backbone.patch_embedding_layer.conv.weight"].reshape(384, 1024).to(dev)
    bb = PG["backbone.patch_embedding_layer.conv.bias"].to(dev)
    x = bag[0].reshape(-1, 1024).to(dev)
    z = ops.linear_act(x, W, bb, "none")
    ln = F.layer_norm(z, (384,), PG["backbone.patch_embedding_layer.norm.weight"].to(dev), PG["backbone.patch_embedding_layer.norm.bias"].to(dev), 1e-5).cpu().double()
    i = (grads["fp64_ln"].abs() == grads["fp64_ln"].abs().min()).nonzero()[0]
    print(f"entry {i.tolist()}: float64 {float(grads['fp64_ln'][tuple(i)]):.3e}, oracle fp32 {float(grads['fp32_ln'][tuple(i)]):.3e}, "
          f"HIP contraction (exact fp32) + LayerNorm {float(ln[tuple(i)]):.3e}")
    flip = (ln > 0) != (grads["fp64_ln"] > 0)
    print("entries on different sides of 0 (HIP fp32 vs float64):", int(flip.sum()), flip.nonzero()[:4].tolist())
