"""GPU: no eager-ATen arithmetic in the product path. Every Linear / LayerNorm of the plugin surface (reference model/backbone.py,
model/model_utils.py, model/GANSurv.py) must reach the HIP library: a step of every backbone, the layer-norm variants of the head
builders (`gen_norm`, `disc_nety_norm`: make_mlp_layer(layer_norm=True), model_utils.py:168-176) and a head whose widths are not
multiples of 4 floats are run with torch.nn.functional.linear / layer_norm replaced by functions that raise."""
import pytest
import torch
import torch.nn.functional as F

from advmil_amd.config import default_cfg
from tests import helpers as H
from tests.test_parity_gpu import DEV, load_synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def no_aten(monkeypatch):
    calls = []

    def boom(name):
        def f(*a, **k):
            calls.append(name)
            raise AssertionError(f"eager torch.nn.functional.{name} reached from the product path")
        return f
    monkeypatch.setattr(F, "linear", boom("linear"))
    monkeypatch.setattr(F, "layer_norm", boom("layer_norm"))
    return calls


def one_epoch(kind, **cfg_over):
    from advmil_amd import synth
    from advmil_amd.model import MyHandler
    lens = (256, 512, 128, 64)
    h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=len(lens), **cfg_over), device=DEV)
    loader = []
    for i, n in enumerate(lens):
        x = H.bag(300 + i, 512)[:, :n].contiguous()
        ext = H.T(synth.cluster_ids(0, 300 + i, n)) if kind == "cluster" else torch.zeros(1, 1)
        loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext], H.label(i)))
    cl = h._train_each_epoch(loader, "train")
    assert bool(torch.isfinite(cl["y_hat"]).all()) and bool(torch.isfinite(cl["f_fake"]).all())
    for p in list(h.netG.parameters()) + list(h.netD.parameters()):
        assert bool(torch.isfinite(p).all())
    return h


@pytest.mark.parametrize("kind", ["abmil", "cluster", "patch"])
def test_a_step_of_every_backbone_issues_no_eager_linear_or_layer_norm(no_aten, kind):
    one_epoch(kind)
    assert no_aten == []


def test_layer_norm_heads_and_odd_widths_stay_on_the_hip_path(no_aten):
    # LayerNorm inside the generator's hop MLP and the discriminator's y-embedding; 62 / 126 are not multiples of 4 floats
    one_epoch("abmil", gen_norm=True, disc_nety_norm=True)
    one_epoch("abmil", disc_nety_hid_dims="62-128")
    assert no_aten == []


def test_the_concat_discriminator_head_stays_on_the_hip_path(no_aten):
    # disc_type = cat (reference model/GANSurv.py:57-75): fc over [hid_x, hid_t] has ONE output column
    one_epoch("abmil", disc_type="cat")
    assert no_aten == []


def test_padded_linear_matches_float64():
    from advmil_amd import ops
    g = torch.Generator().manual_seed(5)
    for (B, K, N) in ((16, 62, 126), (3, 7, 5), (8, 64, 30), (5, 33, 128)):
        x = torch.randn(B, K, generator=g).to(DEV).requires_grad_(True)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
        b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
        go = torch.randn(B, N, generator=g).to(DEV)
        y = ops.linear_act_any(x, W, b, "relu")
        y.backward(go)
        xd, Wd, bd = (t.detach().double().requires_grad_(True) for t in (x, W, b))
        yr = torch.relu(xd @ Wd.t() + bd)
        yr.backward(go.double())
        for got, ref in ((y, yr), (x.grad, xd.grad), (W.grad, Wd.grad), (b.grad, bd.grad)):
            assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (B, K, N)
