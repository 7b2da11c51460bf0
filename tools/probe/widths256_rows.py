"""ESAT d = 256 (head_dim 32), 784 patches, bf16x3 arithmetic: which rows of the first layer's weight gradient leave the oracle's
(tests/test_parity_gpu.py::test_esat_other_backbone_widths_vs_oracle[256] run in bf16x3 mode)? One row = one unit of the region
embedding's LayerNorm -> ReLU whose pre-activation sits on the ReLU boundary."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["ADVMIL_GEMM_MODE"] = "bf16x3"
import torch
from types import SimpleNamespace
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
import tests.test_parity_gpu as T
from tests import helpers as H
from advmil_amd.model import Generator, load_backbone
d = 256
bb = load_backbone("patch", [1024, d, d])
g = Generator(d, 1, bb, SimpleNamespace(noise=[0, 1], hops=1, noise_dist="uniform"), False, 0.6, "sigmoid").to(T.DEV)
PG = T.load_synth(g, f"G-patch{d}:")
T.zero_dropout(g); g.train()
x = H.bag(31, 1024, T.DEV)[:, :784].contiguous()
nz = [H.noise_tensor("esatw", d, d // 2, T.DEV)]
pred = g(x, None, noise=nz); pred.sum().backward()
Pr = {k: v.clone().double().requires_grad_(True) for k, v in PG.items()}
pr = T.O.generator(Pr, x.cpu().double(), None, "patch", (0, 1), [nz[0].cpu().double()], None, "sigmoid")
pr.sum().backward()
k = "backbone.patch_embedding_layer.conv.weight"
got, want = dict(g.named_parameters())[k].grad.cpu().double().reshape(d, -1), Pr[k].grad.reshape(d, -1)
dev = (got - want).abs().max(dim=1)[0]
scale = want.abs().max()
bad = (dev > 1e-3 * scale).nonzero().flatten().tolist()
print("float64 oracle; rows beyond 1e-3 of the tensor's scale:", bad, "their deviations / scale:", [float(dev[i] / scale) for i in bad])
print("largest deviation of the other rows / scale:", float(dev[[i for i in range(d) if i not in bad]].max() / scale))
# pre-activations of the bad units in float64
with torch.no_grad():
    W = Pr[k].detach().reshape(d, -1); b = Pr["backbone.patch_embedding_layer.conv.bias"].detach()
    y = x[0].cpu().double() @ W.t() + b
    ln = torch.nn.functional.layer_norm(y, (d,), Pr["backbone.patch_embedding_layer.norm.weight"].detach(), Pr["backbone.patch_embedding_layer.norm.bias"].detach())
    for j in bad:
        col = ln[:, j]
        i = int(col.abs().argmin())
        print(f"unit {j}: smallest |LayerNorm output| over the 784 rows = {float(col[i]):.3e} (row {i})")
