"""The generator's bag-level head as two launches each way (advmil_ghead_fwd / advmil_ghead_bwd, csrc/ghead.hip; reference
model/GANSurv.py:13-46 behind the backbone's pooling, model/backbone.py:66-70 rho): against a float64 restatement with the kernel's own
dropout masks / noise regenerated on the host, and against the layer-by-layer path it replaces (same call sites, same draws)."""
import numpy as np
import pytest
import torch

from advmil_amd import ops, synth
from advmil_amd.optim import FlatAdam
from tests.test_parity_gpu import DEV, build_generator, load_synth

pytestmark = pytest.mark.gpu


def _run(kind, B, fused, train=True, zero_noise=False, inject=False, noise=(0, 1), seed=11, frozen=False):
    old = ops.GHEAD
    ops.GHEAD = fused
    try:
        g = build_generator(kind)
        if list(noise) != [0, 1]:
            from types import SimpleNamespace
            from advmil_amd.model import Generator, load_backbone
            g = Generator(384, 1, load_backbone(kind, [1024, 384, 384]), SimpleNamespace(noise=list(noise), hops=1, noise_dist="uniform"),
                          False, 0.6, "sigmoid").to(DEV)
            sd = {k: torch.from_numpy(synth.param(7, "Gx:" + k, tuple(v.shape))).to(DEV) for k, v in g.state_dict().items()}
            g.load_state_dict(sd)
        else:
            load_synth(g, f"G-{kind}:")
        g.train(train)
        opt = FlatAdam(g, lr=1e-4)
        opt.zero_grad()
        rng = ops.DeviceRng(DEV, seed=seed)
        rng.record = True
        for m in g.modules():
            m.rng = rng
        gen = torch.Generator().manual_seed(5)
        feats = torch.randn(B, 384, generator=gen).to(DEV).requires_grad_(True)
        nz = [torch.rand(B, 192, generator=gen).to(DEV)] if inject else None
        if frozen:
            for p in g.parameters():
                p.requires_grad_(False)
        pred = g.finish(feats, zero_noise=zero_noise, noise=nz)
        w = torch.linspace(0.5, 1.5, B, device=DEV).reshape(B, 1)
        (pred * w).sum().backward()
        torch.cuda.synchronize()
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in g.named_parameters()}
        return dict(pred=pred.detach(), dfeats=feats.grad, grads=grads, log=list(rng.log), g=g, feats=feats, w=w, nz=nz)
    finally:
        ops.GHEAD = old


def _ref64(r, kind, zero_noise, seed=11):
    g = r["g"]
    P = {k: v.detach().double().cpu().clone().requires_grad_(True) for k, v in g.state_dict().items()}
    x = r["feats"].detach().double().cpu().requires_grad_(True)

    def site(tag):
        ent = [e for e in r["log"] if e[0] == tag]
        return ent[0] if ent else None

    def mask(tag):
        e = site(tag)
        if e is None:
            return None
        _, sid, shape, p = e
        u = synth.kernel_uniform(seed, sid, int(np.prod(shape))).reshape(shape)
        return torch.from_numpy((u >= np.float32(p)).astype(np.float64) / (1 - p))

    h = x
    if kind == "abmil":
        h = torch.relu(h @ P["backbone.rho.0.weight"].t() + P["backbone.rho.0.bias"])
        m = mask("abmil_rho")
        h = h * m if m is not None else h
    h = torch.relu(h @ P["MLPs.0.0.weight"].t() + P["MLPs.0.0.bias"])
    m = mask("gen_mlp0.2")
    h = h * m if m is not None else h
    W1 = P["MLPs.1.0.weight"]
    if W1.shape[1] == 2 * h.shape[1]:
        if zero_noise:
            nz = torch.zeros_like(h)
        elif r["nz"] is not None:
            nz = r["nz"][0].double().cpu()
        else:
            _, sid, shape, _ = site("noise")
            nz = torch.from_numpy(synth.kernel_uniform(seed, sid, int(np.prod(shape))).astype(np.float64)).reshape(h.shape)
        h = torch.cat([h, nz], dim=1)
    pred = torch.sigmoid(h @ W1.t() + P["MLPs.1.0.bias"])
    (pred * r["w"].double().cpu()).sum().backward()
    return pred.detach(), x.grad, {k: v.grad for k, v in P.items()}


def _close(a, b, tol, what):
    if b is None:
        assert a is None or float(a.abs().max()) == 0.0, what
        return
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= tol * scale + 1e-7, (what, float((a - b).abs().max()), scale)


HEAD_KEYS = ("backbone.rho", "MLPs.")


@pytest.mark.parametrize("kind", ["abmil", "patch"])
@pytest.mark.parametrize("B", [1, 3, 16, 32])
def test_fused_head_vs_float64_with_the_kernels_own_masks_and_noise(kind, B):
    r = _run(kind, B, True)
    tags = [e[0] for e in r["log"]]
    assert tags == (["abmil_rho"] if kind == "abmil" else []) + ["gen_mlp0.2", "noise"]
    pred, dx, gp = _ref64(r, kind, False)
    _close(r["pred"], pred, 2e-6, "pred")
    _close(r["dfeats"], dx, 5e-6, "d feats")
    for k, g in r["grads"].items():
        if k.startswith(HEAD_KEYS):
            _close(g, gp[k], 5e-6, k)


@pytest.mark.parametrize("kind", ["abmil", "patch"])
@pytest.mark.parametrize("variant", ["train", "eval", "zero_noise", "inject", "no_noise"])
def test_fused_head_equals_the_layer_by_layer_path(kind, variant):
    kw = dict(train=variant != "eval", zero_noise=variant == "zero_noise", inject=variant == "inject",
              noise=(0, 0) if variant == "no_noise" else (0, 1))
    a, b = _run(kind, 8, True, **kw), _run(kind, 8, False, **kw)
    assert [e[:1] + e[2:] for e in a["log"]] == [e[:1] + e[2:] for e in b["log"]]         # same sites, shapes and rates, in the same order
    assert [e[1] for e in a["log"]] == [e[1] for e in b["log"]]
    _close(a["pred"], b["pred"], 2e-6, "pred")
    _close(a["dfeats"], b["dfeats"], 5e-6, "d feats")
    for k in a["grads"]:
        if k.startswith(HEAD_KEYS):
            assert (a["grads"][k] is None) == (b["grads"][k] is None), k
            if a["grads"][k] is not None:
                _close(a["grads"][k], b["grads"][k], 5e-6, k)


def test_fused_head_with_frozen_parameters_hands_back_the_input_gradient_only():
    a, b = _run("abmil", 4, True, frozen=True), _run("abmil", 4, False, frozen=True)
    _close(a["pred"], b["pred"], 2e-6, "pred")
    _close(a["dfeats"], b["dfeats"], 5e-6, "d feats")
    assert all(g is None or float(g.abs().max()) == 0.0 for g in a["grads"].values())


def test_fused_head_is_what_the_step_runs():
    """The shipped configuration takes the fused launches (a silent fall-back to the layer-by-layer path would pass every parity test)."""
    g = build_generator("abmil").train()
    FlatAdam(g, lr=1e-4).zero_grad()
    feats = torch.randn(16, 384, device=DEV)
    assert g._head_spec(feats, False, None) is not None
