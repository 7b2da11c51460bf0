#!/usr/bin/env python3
"""Raw gradients (one D backward, one G backward, no Adam, dropout off) of one bag geometry against the FLOAT64 oracle, every parameter
printed (tests/test_handler_variants_gpu.py::gradients_before_adam asserts the first one that is off). Device bags: no staging slab, no pad.
usage: grad_case_check.py <backbone> <bag seed> <len> [<len> ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import synth  # noqa: E402
from advmil_amd.config import default_cfg  # noqa: E402
from advmil_amd.model import MyHandler  # noqa: E402
from oracle import advmil_oracle as O  # noqa: E402   (a probe: test infrastructure, like tests/)
from tests import helpers as H  # noqa: E402
from tests.test_parity_gpu import DEV, load_synth, zero_dropout  # noqa: E402

kind, seed0, lens = sys.argv[1], int(sys.argv[2]), [int(v) for v in sys.argv[3:]]
nb = len(lens)
for mode in os.environ.get("GC_MODES", "exact,bf16x3").split(","):
    h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb, gemm_mode=mode), device=DEV)
    PG, PD = load_synth(h.netG, f"G-{kind}:"), load_synth(h.netD, "D-prj:")
    zero_dropout(h.netG); zero_dropout(h.netD)
    bags = []
    for i, n in enumerate(lens):
        ext = H.T(synth.cluster_ids(0, seed0 + i, n)) if kind == "cluster" else None
        bags.append((H.bag(seed0 + i, max(512, max(lens)))[:, :n].contiguous(), ext, H.label(i)))
    xs = [[b[0].to(DEV), b[1].to(DEV) if b[1] is not None else torch.zeros(1, 1, device=DEV)] for b in bags]
    ys_host = [b[2] for b in bags]
    ys = [y.to(DEV) for y in ys_host]
    nd = [[H.noise_tensor("gr_d", i, 192)] for i in range(nb)]
    ng = [[H.noise_tensor("gr_g", i, 192)] for i in range(nb)]
    plan = h._plan(xs, ys, "wlabel", None, ys_host)
    h._disc_backward(0, xs, ys, plan, [[n[0].to(DEV)] for n in nd])
    h._gen_backward(0, xs, ys, plan, [[n[0].to(DEV)] for n in ng])
    torch.cuda.synchronize()
    cfg = O.StepConfig(kind=kind, loss_netD="bce", l1_coef=0.0, disc_type="prj", inner_product="instance", prj_path="x")
    dbl = lambda d: {k_: v_.double() for k_, v_ in d.items()}  # noqa: E731
    bags64 = [(x_.double(), None if e_ is None else e_.double(), y_.double()) for x_, e_, y_ in bags]
    _, gD, _, _ = O.update_disc(cfg, dbl(PG), dbl(PD), bags64, [[n[0].double()] for n in nd])
    _, gG, _ = O.update_gen(cfg, dbl(PG), dbl(PD), bags64, [[n[0].double()] for n in ng])
    print(f"== {mode}: {kind}, {sum(lens)} rows (mod 256: {sum(lens) % 256})")
    for tag, net, want in (("D", h.netD, gD), ("G", h.netG, gG)):
        for k, p in net.named_parameters():
            w = want.get(k)
            if w is None:
                continue
            got = torch.zeros_like(p) if p.grad is None else p.grad
            scale = float(w.abs().max())
            err = float((got.cpu().double() - w).abs().max())
            off = err > 2e-5 * scale + 2.5e-7
            if off and got.dim() >= 2:
                rows_err = (got.cpu().double() - w).abs().reshape(got.shape[0], -1).max(dim=1).values
                top = torch.topk(rows_err, min(3, rows_err.numel()))
                k = k + " [units " + ",".join(str(int(i)) for i in top.indices) + "]"
            if off or not os.environ.get("GC_OFF_ONLY"):
                print(f"   {tag} {k:44s} max err {err:.2e} of scale {scale:.2e}  rel {err / (scale + 1e-30):.1e}" + ("   <-- OFF" if off else ""))

# ---- are the deviating gradient rows the units whose float64 pre-activation lies within fp32 round-off of the ReLU boundary?
# (such an entry takes either branch in ANY fp32 evaluation, depending on the summation order; the forward does not move, the unit's weight /
# bias gradient row and everything upstream of it does: DESIGN.md section 2)
if kind == "cluster":
    P = {k: v.double() for k, v in PG.items()}
    W1 = P["backbone.phis.0.weight"].reshape(P["backbone.phis.0.weight"].shape[0], -1)
    b1 = P["backbone.phis.0.bias"]
    W2, b2 = P["backbone.attention_net.0.weight"], P["backbone.attention_net.0.bias"]
    near1, near2 = [], []
    row0 = 0
    for bi_, (x_, e_, _) in enumerate(bags):
        z = x_[0].double() @ W1.t() + b1
        for r, c in (z.abs() < 2e-6).nonzero().tolist():
            near1.append((row0 + r, c, float(z[r, c])))
        hrel = torch.relu(z)
        ids = e_.reshape(-1).long()
        m = torch.stack([hrel[ids == c].mean(dim=0) if bool((ids == c).any()) else torch.zeros(hrel.shape[1], dtype=torch.float64) for c in range(8)])
        z2 = m @ W2.t() + b2
        for r, c in (z2.abs() < 2e-6).nonzero().tolist():
            near2.append((bi_, r, c, float(z2[r, c])))
        row0 += x_.shape[1]
    print("float64 pre-activations within 2e-6 of the ReLU boundary: patch level (row, unit, value):", near1[:8])
    print("                                                          cluster level (bag, cluster, unit, value):", near2[:8])
