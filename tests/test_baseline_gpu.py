"""GPU: the supervised baselines against golden G7 -- two optimizer steps through the REFERENCE's own
BaselineHandler._update_network (tests/golden/gen_golden_baseline.py) -- and against the oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from advmil_amd import synth
from advmil_amd.config import default_baseline_cfg
from oracle import advmil_oracle as O
from tests import helpers as H
from tests.test_parity_gpu import DEV, load_synth, zero_dropout

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_baseline_v1.npz"))
CONFIGS = [("abmil_reg", "abmil", "surv_reg", "384-1"), ("cluster_cox", "cluster", "surv_cox", "384-1"),
           ("abmil_nll", "abmil", "surv_nll", "384-4"), ("patch_reg", "patch", "surv_reg", "384-1")]
NB, N, STEPS = 8, 512, 2


def labels(task, i):
    y = H.label(i).clone()
    if task == "surv_nll":
        y[0, 0] = float(int(y[0, 0] * 4) % 4)
    elif task == "surv_cox":
        y[0, 0] = y[0, 0] * 100.0 + i * 0.01
    return y


def run(tag, kind, task, pdh, gemm_mode):
    from advmil_amd import ops
    from advmil_amd.model import BaselineHandler
    prev = ops.get_gemm_mode()
    try:
        h = BaselineHandler(default_baseline_cfg(bcb_mode=kind, task=task, pdh_dims=pdh, bp_every_batch=NB, gemm_mode=gemm_mode), device=DEV)
        P0 = load_synth(h.net, f"S-{tag}:")
        zero_dropout(h.net)
        loader = []
        for i in range(STEPS * NB):
            x = H.bag(100 + i, N)
            ext = H.T(synth.cluster_ids(0, 100 + i, N)) if kind == "cluster" else torch.zeros(1, 1)
            loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext], labels(task, i)))
        cl = h._train_each_epoch(loader, "train")
        return h, P0, cl, h.pop_logs()
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("gemm_mode", ["exact", "bf16x3"])
@pytest.mark.parametrize("tag,kind,task,pdh", CONFIGS)
def test_two_optimizer_steps_vs_reference(tag, kind, task, pdh, gemm_mode):
    h, P0, cl, logs = run(tag, kind, task, pdh, gemm_mode)
    want_logs, want_preds = GOLD[f"G7_{tag}_logs"], GOLD[f"G7_{tag}_preds"]
    for s in range(STEPS):
        assert abs(logs[s]["train_batch/net/loss_supervision"] - want_logs[s, 0]) < 2e-5, (s, logs[s], want_logs[s])
        assert abs(logs[s]["train_batch/net/loss_total"] - want_logs[s, 1]) < 2e-5, (s, logs[s], want_logs[s])
        got = cl["y_hat"][s * NB:(s + 1) * NB].numpy()
        assert np.abs(got - want_preds[s]).max() < 2e-5, (s, np.abs(got - want_preds[s]).max())
    # post-step weights: per-tensor sum / norm of the two-step delta (Adam amplifies round-off on noise-level gradients)
    keys = [str(k) for k in GOLD[f"G7_{tag}_keys"]]
    sd = h.net.state_dict()
    for k, (dsum, dnorm) in zip(keys, GOLD[f"G7_{tag}_dstats"]):
        d = (sd[k].double().cpu() - P0[k].double())
        assert abs(float(d.norm()) - dnorm) < 2e-2 * dnorm + 5e-6, (k, float(d.norm()), dnorm)


def test_losses_equal_oracle():
    from advmil_amd.loss.utils import MSE_loss, SurvMLE, SurvPLE
    g = torch.Generator().manual_seed(4)
    t, e = torch.rand(12, 1, generator=g), (torch.rand(12, 1, generator=g) < 0.5).float()
    p = torch.rand(12, 1, generator=g)
    for inc in (False, True):
        assert abs(float(MSE_loss(p.to(DEV), t.to(DEV), e.to(DEV), inc)) - float(O.mse_loss(p, t, e, inc))) < 1e-6
    hz = torch.rand(12, 4, generator=g) * 0.9 + 0.05
    tb = torch.randint(0, 4, (12, 1), generator=g).float()
    for a in (0.0, 0.3):
        assert abs(float(SurvMLE(alpha=a)(hz.to(DEV), tb.to(DEV), e.to(DEV))) - float(O.surv_mle(hz, tb, e, a))) < 1e-6
    th = torch.randn(12, 1, generator=g) * 3 + 8        # some above the cap of 10
    assert abs(float(SurvPLE()(th.to(DEV), (t * 50).to(DEV), e.to(DEV))) - float(O.surv_ple(th, t * 50, e))) < 1e-5


def test_test_model_and_checkpoint_round_trip(tmp_path):
    from advmil_amd.model import BaselineHandler
    h = BaselineHandler(default_baseline_cfg(bcb_mode="abmil", task="surv_reg", bp_every_batch=2, save_path=str(tmp_path)), device=DEV)
    loader = [(torch.tensor([[i]], dtype=torch.int), [H.bag(300 + i, 256), torch.zeros(1, 1)], H.label(i)) for i in range(4)]
    h._train_each_epoch(loader, "train")
    h.save_model(1, "last")
    a = BaselineHandler.test_model(h.net, "abmil", loader, times_test_sample=3)
    assert a["y_hat"].shape == (4, 1) and a["dist_y_hat"].shape == (4, 3, 1) and torch.equal(a["avg_y_hat"], a["y_hat"])
    h2 = BaselineHandler(default_baseline_cfg(bcb_mode="abmil", task="surv_reg", bp_every_batch=2, save_path=str(tmp_path)), device=DEV)
    h2.resume_model("last")
    b = BaselineHandler.test_model(h2.net, "abmil", loader)
    assert torch.equal(a["y_hat"], b["y_hat"])


@pytest.mark.parametrize("kind", ["abmil", "cluster", "patch"])
def test_batched_eval_and_staged_training_equal_the_per_bag_forms(kind):
    """The shared ingest (advmil_amd/ingest.py::step_batches): host bags through the staging slab == the same bags handed over as
    device tensors (training, two steps); evaluation in slabs of 3 == the per-bag loop (ragged bags)."""
    from advmil_amd.model import BaselineHandler
    lens = (256, 128, 512, 64, 192, 384, 320, 96)

    def loader(device=None):
        out = []
        for i, n in enumerate(lens):
            x = H.bag(500 + i, 512)[:, :n].contiguous()
            ext = H.T(synth.cluster_ids(0, 500 + i, n)) if kind == "cluster" else torch.zeros(1, 1)
            if device is not None:
                x, ext = x.to(device), ext.to(device)
            out.append((torch.tensor([[i]], dtype=torch.int), [x, ext], H.label(i)))
        return out

    def train(ld):
        h = BaselineHandler(default_baseline_cfg(bcb_mode=kind, task="surv_reg", bp_every_batch=3), device=DEV)
        load_synth(h.net, f"S-{kind}-ev:")
        zero_dropout(h.net)
        return h, h._train_each_epoch(ld, "train"), h.pop_logs()

    (ha, cla, la), (hb, clb, lb) = train(loader()), train(loader(DEV))
    assert cla["y_hat"].shape == (6, 1) and len(la) == len(lb) == 2          # 8 bags, steps of 3: the trailing two are dropped
    assert float((cla["y_hat"] - clb["y_hat"]).abs().max()) <= 2e-6 and torch.equal(cla["y"], clb["y"])
    for a, b in zip(la, lb):
        assert abs(a["train_batch/net/loss_supervision"] - b["train_batch/net/loss_supervision"]) <= 2e-6
    one = BaselineHandler.test_model(ha.net, kind, loader(), times_test_sample=2, batch_bags=1)
    bat = BaselineHandler.test_model(ha.net, kind, loader(), times_test_sample=2, batch_bags=3)
    assert set(one) == set(bat) == {"idx", "y", "y_hat", "dist_y_hat", "avg_y_hat"}
    for k in one:
        assert one[k].shape == bat[k].shape and float((one[k].double() - bat[k].double()).abs().max()) <= 2e-6, k
