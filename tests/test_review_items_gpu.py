"""Regression cases for review findings: the configured bag storage reaches the evaluation passes, the no-grad GENConv pass writes `out`
only, the [B <= 32, d] fp32-FMA layers at their 16- / 32-row instantiations."""
import pytest
import torch

from advmil_amd import ops
from tests import helpers as H
from tests.test_parity_gpu import DEV, make_handler

pytestmark = pytest.mark.gpu


def test_cfg_only_bf16_bag_storage_reaches_the_evaluation_pass(monkeypatch):
    """cfg['x_storage'] = 'bf16' with no environment switch: test_model (static, reference signature) must stage its slabs in bf16 too --
    the same rounded bags, the same device-cache entries the training loop uses (model_handler.py:598-643 evaluates what it trained on)."""
    from advmil_amd.model import MyHandler
    monkeypatch.delenv("ADVMIL_X_STORAGE", raising=False)
    h, _, _ = make_handler("abmil", bp_every_batch=2, x_storage="bf16", gemm_mode="bf16x3")
    seen = []
    real = MyHandler._slab_build_static

    def spy(xs, resident_planes=True, pad=0):
        X = real(xs, resident_planes, pad)
        seen.append(X.dtype)
        return X
    monkeypatch.setattr(MyHandler, "_slab_build_static", staticmethod(spy))
    loader = [(torch.tensor([[i]], dtype=torch.int), [H.bag(i, 512), torch.zeros(1, 1)], H.label(i)) for i in range(4)]
    res = MyHandler.test_model(h.netG, h.netD, "abmil", loader, times_test_sample=1, test_zero_noise=True)
    assert res["y_hat"].shape[0] == 4 and seen and all(dt == torch.bfloat16 for dt in seen), seen
    seen.clear()
    h.patient_id["label_visible"] = h.patient_id["train"] = [str(i) for i in range(4)]
    h._train_each_epoch(loader, "train")
    ops.set_gemm_mode("exact")


def test_no_grad_genconv_pass_keeps_nothing_for_a_backward(monkeypatch):
    """GenConvAggFn under torch.no_grad(): the temperature is a Parameter (needs_input_grad says True whatever the grad mode), yet the
    evaluation pass must hand the kernel NULL lse / agg rows (`out` only)."""
    from types import SimpleNamespace
    from advmil_amd import synth
    n = 256
    x = torch.randn(n, 128, device=DEV, requires_grad=True)
    t = torch.nn.Parameter(torch.ones(1, device=DEV))
    csr = ops.graph_csr(SimpleNamespace(x=x.detach(), edge_index=H.T(synth.grid_knn_graph(n, 8), DEV).long()))
    L = ops._lib.lib()
    real = L.advmil_genconv_fwd
    saved = []

    def spy(*a):
        saved.append((a[8], a[9]))           # lse, agg
        return real(*a)
    monkeypatch.setattr(L, "advmil_genconv_fwd", spy)
    out_g = ops.genconv_aggregate(x, t, csr)
    with torch.no_grad():
        out_n = ops.genconv_aggregate(x, t, csr)
    torch.cuda.synchronize()
    ptr = lambda v: None if v is None else getattr(v, "value", v)
    assert ptr(saved[0][0]) and ptr(saved[0][1])                 # the training pass keeps both rows
    assert not ptr(saved[1][0]) and not ptr(saved[1][1])         # the no-grad pass keeps neither
    assert out_g.grad_fn is not None and out_n.grad_fn is None
    err = float((out_g.detach().double() - out_n.double()).abs().max())
    assert err <= 1e-6 * float(out_n.abs().max()), err


@pytest.mark.parametrize("M", [16, 32])
def test_small_linear_kernels_at_16_and_32_rows(M, monkeypatch):
    monkeypatch.setattr(ops, "SMALL_LINEAR_ROWS", 32)
    g = torch.Generator().manual_seed(M)
    K, N = 384, 192
    x = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
    W = (0.05 * torch.randn(N, K, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(N, generator=g)).to(DEV).requires_grad_(True)
    y = ops.linear_act(x, W, b, "relu")
    (y * torch.linspace(0.5, 1.5, N, device=DEV)).sum().backward()
    xr, Wr, br = (v.detach().double().cpu().requires_grad_(True) for v in (x, W, b))
    yr = torch.relu(xr @ Wr.t() + br)
    (yr * torch.linspace(0.5, 1.5, N).double()).sum().backward()
    for a, r, nm in ((y, yr, "y"), (x.grad, xr.grad, "dx"), (W.grad, Wr.grad, "dW"), (b.grad, br.grad, "db")):
        scale = float(r.abs().max()) + 1e-12
        assert float((a.detach().double().cpu() - r.detach()).abs().max()) <= 2e-6 * scale + 1e-7, nm
