"""The discriminator's region-level network as one launch each way (advmil_dx_chain_fwd / _bwd, csrc/region.hip; reference
model/model_utils.py:188-210 EmbedXLayer.fc1 + model/backbone_utils.py:31-56 GAPool): against the layer-by-layer path it replaces -- same
call sites, same dropout draws, same bf16x3 products -- and against float64 for the forward."""
import numpy as np
import pytest
import torch

from advmil_amd import ops
from advmil_amd.optim import FlatAdam
from tests.test_parity_gpu import DEV, build_disc, load_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16x3():
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    yield
    ops.set_gemm_mode(prev)


def _run(lens, fused, train=True, want_mean=True, seed=7, e_grad=True):
    old = ops.DX_CHAIN
    ops.DX_CHAIN = fused
    try:
        d = build_disc("prj", "instance", "x")
        d.train(train)
        load_synth(d, "D-prj:")
        opt = FlatAdam(d, lr=1e-4)
        opt.zero_grad()
        rng = ops.DeviceRng(DEV, seed=seed)
        rng.record = True
        for m in d.modules():
            m.rng = rng
        R = sum(lens)
        g = torch.Generator().manual_seed(5)
        e = torch.randn(R, 128, generator=g).to(DEV).requires_grad_(e_grad)
        seg = ops.Segments(lens, DEV) if len(lens) > 1 else None
        out = d.net_pair_one.pool_features_rows(e, seg, want_mean=want_mean)
        pooled, fc = out[0], out[1]
        mean = out[2] if want_mean else None
        nb = len(lens)
        w1 = torch.randn(nb, 128, generator=g).to(DEV)
        w2 = torch.randn(nb, 128, generator=g).to(DEV)
        loss = (pooled.reshape(nb, 128) * w1).sum()
        if mean is not None:
            loss = loss + (mean.reshape(nb, 128) * w2).sum()
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in d.named_parameters() if k.startswith("net_pair_one.fc1") or k.startswith("net_pair_one.pool")}
        return dict(pooled=pooled.detach(), fc=fc.detach(), mean=None if mean is None else mean.detach(), de=e.grad, grads=grads,
                    log=[(t, s, sh, p) for (t, s, sh, p) in rng.log], A=d.net_pair_one.pool.last_attention.clone(), d=d, e=e.detach())
    finally:
        ops.DX_CHAIN = old


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, what
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    # (absolute floor: the scorer's output bias has a TRUE gradient of zero under the softmax -- sum of ds over a bag --, both paths hold
    # only round-off there, ~1e-6)
    floor = 5e-6 if what.endswith("pool.fc2.bias") else 1e-7
    assert err <= tol * scale + floor, (what, err, scale)


@pytest.mark.parametrize("lens", [[64], [37], [512, 96, 130], [2048] * 4, [1, 300]])
@pytest.mark.parametrize("train", [True, False])
def test_fused_region_network_equals_the_layer_by_layer_path(lens, train):
    a, b = _run(lens, True, train), _run(lens, False, train)
    assert a["log"] == b["log"]                         # same call sites, stream ids, shapes and rates, in the same order
    assert train == any(t == "dx_fc1" for (t, _, _, _) in a["log"])
    _close(a["fc"], b["fc"], 1e-6, "fc_ins")
    _close(a["A"], b["A"], 2e-5, "attention weights")
    _close(a["pooled"], b["pooled"], 1e-5, "pooled")
    _close(a["mean"], b["mean"], 1e-5, "mean")
    _close(a["de"], b["de"], 5e-5, "d e")
    for k in a["grads"]:
        _close(a["grads"][k], b["grads"][k], 5e-5, k)


def test_fused_region_forward_vs_float64():
    lens = [700, 20, 336]
    r = _run(lens, True, train=False)
    P = {k: v.detach().double().cpu() for k, v in r["d"].state_dict().items()}
    e = r["e"].double().cpu()
    p = "net_pair_one."
    h1 = torch.relu(e @ P[p + "fc1.0.weight"].t() + P[p + "fc1.0.bias"])
    fc = h1 @ P[p + "fc1.3.weight"].t() + P[p + "fc1.3.bias"]
    a = torch.tanh(fc @ P[p + "pool.fc1.0.weight"].t() + P[p + "pool.fc1.0.bias"])
    b = torch.sigmoid(fc @ P[p + "pool.score.0.weight"].t() + P[p + "pool.score.0.bias"])
    s = (a * b) @ P[p + "pool.fc2.weight"].t() + P[p + "pool.fc2.bias"]
    _close(r["fc"], fc, 2e-5, "fc_ins")
    o, pooled, mean = 0, [], []
    for n in lens:
        A = torch.softmax(s[o:o + n, 0], dim=0)
        _close(r["A"][o:o + n], A, 1e-4, "A")
        pooled.append(A @ fc[o:o + n]); mean.append(fc[o:o + n].mean(dim=0))
        o += n
    _close(r["pooled"], torch.stack(pooled), 5e-5, "pooled")
    _close(r["mean"], torch.stack(mean), 5e-5, "mean")


def test_fused_region_network_without_the_mean_and_without_an_input_gradient():
    a, b = _run([256, 64], True, want_mean=False, e_grad=False), _run([256, 64], False, want_mean=False, e_grad=False)
    _close(a["pooled"], b["pooled"], 1e-5, "pooled")
    assert a["de"] is None
    for k in a["grads"]:
        _close(a["grads"][k], b["grads"][k], 5e-5, k)


def test_pooling_with_the_mean_from_the_same_pass():
    g = torch.Generator().manual_seed(1)
    lens = [513, 1, 4096, 77]
    N = sum(lens)
    h = torch.randn(N, 128, generator=g).to(DEV)
    s = (3.0 * torch.randn(N, generator=g)).to(DEV)
    seg = ops.Segments(lens, DEV)
    A0, p0 = ops.softmax_pool(s, h, N, 128, seg)
    A1, p1, m1 = ops.softmax_pool_mean(s, h, N, 128, seg)
    assert torch.equal(A0, A1) and torch.equal(p0, p1)
    o = 0
    for i, n in enumerate(lens):
        _close(m1[i], h[o:o + n].double().mean(dim=0), 1e-5, f"mean {i}")
        o += n


def test_region_embedding_duplicated_by_its_kernel_equals_the_concatenation():
    """embed_rows(X, dup = 2) = [emb; emb] straight from the LayerNorm kernel, its backward summing the halves of the gradient on load,
    against torch.cat([emb, emb]) + autograd's add."""
    res = {}
    for dup_on in (True, False):
        import os
        os.environ["ADVMIL_LN_DUP"] = "1" if dup_on else "0"
        try:
            d = build_disc("prj", "instance", "x").train()
            load_synth(d, "D-prj:")
            opt = FlatAdam(d, lr=1e-4)
            opt.zero_grad()
            g = torch.Generator().manual_seed(4)
            X = torch.randn(4096 + 512, 1024, generator=g).to(DEV)
            e2 = d.embed_rows(X, 2)
            L = X.shape[0] // 16
            assert e2.shape == (2 * L, 128) and torch.equal(e2[:L], e2[L:])
            w = torch.randn(2 * L, 128, generator=g).to(DEV)
            (e2 * w).sum().backward()
            torch.cuda.synchronize()
            res[dup_on] = (e2.detach().clone(), {k: p.grad.clone() for k, p in d.named_parameters() if "embedding" in k})
        finally:
            os.environ.pop("ADVMIL_LN_DUP", None)
    assert torch.equal(res[True][0], res[False][0])
    for k in res[True][1]:
        _close(res[True][1][k], res[False][1][k], 2e-5, k)
