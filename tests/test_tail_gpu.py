"""The discriminator's bag-level tail as one launch each way (advmil_dtail_fwd / advmil_dtail_bwd, csrc/tail.hip; reference
model/GANSurv.py:89-105): against a float64 restatement with the kernel's own dropout masks regenerated on the host, and against the
layer-by-layer path it replaces (same call sites, same draws)."""
import numpy as np
import pytest
import torch

from advmil_amd import ops, synth
from advmil_amd.optim import FlatAdam
from tests import helpers as H
from tests.test_parity_gpu import DEV, build_disc, load_synth

pytestmark = pytest.mark.gpu


def _run(iprd, prj, B, fused, p_y=0.3, frozen=False, seed=11, thaw=False):
    old = ops.DTAIL
    ops.DTAIL = fused
    try:
        d = build_disc("prj", iprd, prj).train()
        load_synth(d, "D-prj:")
        for m in d.net_pair_two.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = p_y
        opt = FlatAdam(d, lr=1e-4)                     # parameters -> arena (the fused backward adds its weight gradients in place)
        opt.zero_grad()
        rng = ops.DeviceRng(DEV, seed=seed)
        rng.record = True
        for m in d.modules():
            m.rng = rng
        g = torch.Generator().manual_seed(3)
        eb = torch.randn(B, 128, generator=g).to(DEV).requires_grad_(not frozen)
        im = torch.randn(B, 128, generator=g).to(DEV).requires_grad_(not frozen)
        t = torch.rand(B, 1, generator=g).to(DEV).requires_grad_(True)
        if frozen:
            for p in d.parameters():
                p.requires_grad_(False)
        f = d.tail(eb, im if iprd == "instance" else None, t)
        if frozen and thaw:                  # the handler's generator update: the flags are back on before backward() runs (model_handler.py:409)
            for p in d.parameters():
                p.requires_grad_(True)
        w = torch.linspace(0.5, 1.5, B, device=DEV).reshape(B, 1)
        (f * w).sum().backward()
        torch.cuda.synchronize()
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in d.named_parameters()}
        return dict(f=f.detach(), deb=eb.grad, dim=im.grad, dt=t.grad, grads=grads, log=list(rng.log), d=d, eb=eb, im=im, t=t, w=w)
    finally:
        ops.DTAIL = old


def _ref64(r, iprd, prj, seed=11):
    """float64 restatement with the recorded call sites' masks."""
    d = r["d"]
    P = {k: v.detach().double().cpu() for k, v in d.state_dict().items()}
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    eb, im, t = (r[k].detach().double().cpu().requires_grad_(True) for k in ("eb", "im", "t"))

    def mask(tag):
        ent = [e for e in r["log"] if e[0] == tag]
        if not ent:
            return None
        _, sid, shape, p = ent[0]
        u = synth.kernel_uniform(seed, sid, int(np.prod(shape))).reshape(shape)
        return torch.from_numpy((u >= np.float32(p)).astype(np.float64) / (1 - p))

    def lin(x, w, b):
        return x @ P[w].t() + P[b]
    h1 = torch.relu(lin(eb, "net_pair_one.fc2.0.weight", "net_pair_one.fc2.0.bias"))
    m = mask("dx_fc2.2")
    h1 = h1 * m if m is not None else h1
    hx = lin(h1, "net_pair_one.fc2.3.weight", "net_pair_one.fc2.3.bias")
    t1 = torch.relu(lin(t, "net_pair_two.0.0.weight", "net_pair_two.0.0.bias"))
    m = mask("dy.0.2")
    t1 = t1 * m if m is not None else t1
    ht = torch.relu(lin(t1, "net_pair_two.1.0.weight", "net_pair_two.1.0.bias"))
    m = mask("dy.1.2")
    ht = ht * m if m is not None else ht
    u = im if iprd == "instance" else hx
    out = (u * ht).sum(dim=1, keepdim=True)
    if prj is not None:
        out = out + lin(hx if prj == "x" else ht, "prj_layer.weight", "prj_layer.bias")
    (out * r["w"].double().cpu()).sum().backward()
    return out.detach(), eb.grad, im.grad, t.grad, {k: v.grad for k, v in P.items()}


def _close(a, b, tol, what):
    if b is None:                            # the float64 graph never touched this input: the kernel must hand back zeros (or nothing)
        assert a is None or float(a.abs().max()) == 0.0, what
        return
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= tol * scale + 1e-7, (what, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("iprd,prj", [("instance", "x"), ("bag", "x"), ("instance", "y"), ("bag", None)])
@pytest.mark.parametrize("B", [1, 4, 32])
def test_fused_tail_vs_float64_with_the_kernels_own_masks(iprd, prj, B):
    r = _run(iprd, prj, B, True)
    assert any(e[0] == "dx_fc2.2" for e in r["log"]) and any(e[0] == "dy.1.2" for e in r["log"])
    out, deb, dim_, dt, gp = _ref64(r, iprd, prj)
    _close(r["f"], out, 2e-6, "f")
    _close(r["deb"], deb, 5e-6, "d emb_bag")
    _close(r["dt"], dt, 5e-6, "d t")
    if iprd == "instance":
        _close(r["dim"], dim_, 5e-6, "d ins_mean")
    for k, g in r["grads"].items():
        if gp.get(k) is None or not (k.startswith("net_pair_one.fc2") or k.startswith("net_pair_two") or k.startswith("prj_layer")):
            continue
        _close(g, gp[k], 5e-6, k)


@pytest.mark.parametrize("iprd,prj", [("instance", "x"), ("bag", "x")])
def test_fused_tail_equals_the_layer_by_layer_path(iprd, prj):
    a, b = _run(iprd, prj, 8, True), _run(iprd, prj, 8, False)
    assert [e[:1] + e[2:] for e in a["log"]] == [e[:1] + e[2:] for e in b["log"]]         # same sites, shapes and rates, in the same order
    assert [e[1] for e in a["log"]] == [e[1] for e in b["log"]]
    _close(a["f"], b["f"], 2e-6, "f")
    for k in ("deb", "dt"):
        _close(a[k], b[k], 5e-6, k)
    for k in a["grads"]:
        if a["grads"][k] is not None and b["grads"][k] is not None and float(b["grads"][k].abs().max()) > 0:
            _close(a["grads"][k], b["grads"][k], 5e-6, k)


@pytest.mark.parametrize("thaw", [False, True])
def test_fused_tail_with_frozen_discriminator_returns_only_the_label_gradient(thaw):
    """The generator update: D's parameters are frozen around the FORWARD, nothing of D(x) is differentiated; only d f / d pred leaves the
    tail -- also when requires_grad is back on by the time the backward runs."""
    a, b = _run("instance", "x", 16, True, frozen=True, thaw=thaw), _run("instance", "x", 16, False, frozen=True, thaw=thaw)
    _close(a["f"], b["f"], 2e-6, "f")
    _close(a["dt"], b["dt"], 5e-6, "dt")
    assert a["deb"] is None and all(g is None or float(g.abs().max()) == 0.0 for g in a["grads"].values())
