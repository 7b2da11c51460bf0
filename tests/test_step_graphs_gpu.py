"""Shape-keyed step graphs of the epoch loop (model_handler.StaticStepPlan, round 6): a resident RAGGED step batch whose key -- bags per
step, padded slab rows, number of real pairs / visible labels, slab buffer -- was seen before is replayed from a HIP graph whose plan
arrays (segment offsets, per-row bag ids, labels, masks, dropout row map) are rewritten for that batch, instead of re-issuing ~53 eager
launches (the reference re-issues ~480 per bag: model_handler.py:311-345). Replaying must not move a bit: the same epochs with
no graph ever captured (cfg step_graphs_max = 0: every step eager, same pad and launch grids) give the same weights, predictions, scores
and logs -- shipped dropout ON."""
import pytest
import torch

from advmil_amd.config import default_cfg
from tests import helpers as H
from tests.test_parity_gpu import DEV, load_synth

pytestmark = pytest.mark.gpu

# 10 step batches of 4 bags: every batch has 8192 rows in all and a longest bag in (2048, 4096] (one key per slab buffer) but another split
# into bags; 2 events per batch in changing positions
LENS = [(2048, 1024, 3072, 2048), (1024, 2048, 2048, 3072), (4096, 1024, 1024, 2048), (2560, 2560, 1536, 1536), (512, 3584, 2048, 2048),
        (2048, 2560, 1536, 2048), (3072, 3072, 1024, 1024), (1536, 2560, 3072, 1024), (1024, 1024, 2048, 4096), (2048, 3072, 1024, 2048)]
EVENTS = [(1, 0, 1, 0), (0, 1, 1, 0), (1, 1, 0, 0), (0, 0, 1, 1), (1, 0, 0, 1), (0, 1, 0, 1), (1, 0, 1, 0), (1, 1, 0, 0), (0, 1, 1, 0), (0, 0, 1, 1)]


def _epochs(graphs, epochs=2, lens=LENS, odd=False):
    from advmil_amd import synth
    from advmil_amd.model import MyHandler
    h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=4, gemm_mode="bf16x3", step_graphs=1, step_graphs_max=24 if graphs else 0, bag_cache_gb=2), device=DEV)
    load_synth(h.netG, "G-abmil:"); load_synth(h.netD, "D-prj:")
    h.optimizerG.refresh_planes(); h.optimizerD.refresh_planes()
    h.rng.reset(77)
    n = 4 * len(lens)
    h.patient_id.update({"train": [f"t{i}" for i in range(n)], "label_visible": [f"t{i}" for i in range(n)]})
    loader, i = [], 0
    for bl, be in zip(lens, EVENTS):
        for m, e in zip(bl, be):
            m = m - 16 if (odd and i % 3 == 0) else m                 # rows that are a multiple of 16 but not of 256: the slab pad is in play
            x = H.T(synth.bag(H.DATA_SEED, 400 + i, m)).pin_memory()
            loader.append((torch.tensor([[i]], dtype=torch.int), [x, torch.zeros(1, 1)], torch.tensor([[0.2 + 0.013 * i, float(e)]])))
            i += 1
    cls = [h._train_each_epoch(loader, "train", "wlabel") for _ in range(epochs)]
    torch.cuda.synchronize()
    return h, cls, h.pop_logs()


@pytest.mark.parametrize("odd", [False, True])
def test_replayed_ragged_steps_equal_eager_steps(odd):
    from advmil_amd import ops
    prev = ops.get_gemm_mode()
    try:
        hg, cg, lg = _epochs(True, odd=odd)
        he, ce, le = _epochs(False, odd=odd)
    finally:
        ops.set_gemm_mode(prev)
    # a key per slab buffer and staging kind (epoch 1: bags over PCIe, rows + planes staged; epoch 2: cached bags staged as planes only);
    # 2 eager-or-capture sightings per key, every later batch replayed; nothing with the switch off
    assert len(hg._step_graph_cache) == 4 and len(he._step_graph_cache) == 0
    assert hg.step_graph_stats == {"replayed": 12, "captured": 4, "eager": 4}, hg.step_graph_stats
    assert he.step_graph_stats == {"replayed": 0, "captured": 0, "eager": 20}, he.step_graph_stats
    for a, b in zip(cg, ce):
        for k in ("y", "y_hat", "f_fake"):
            assert torch.equal(a[k], b[k]), k
    assert len(lg) == len(le) == 2 * 10 * 2
    for a, b in zip(lg, le):
        assert a.keys() == b.keys()
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])
    assert torch.equal(hg.optimizerG.flat_param, he.optimizerG.flat_param) and torch.equal(hg.optimizerD.flat_param, he.optimizerD.flat_param)
    assert int(hg.optimizerG.step_t.item()) == int(he.optimizerG.step_t.item()) == 20


def test_a_new_key_falls_back_to_eager_and_a_changed_learning_rate_recaptures():
    from advmil_amd import ops
    prev = ops.get_gemm_mode()
    try:
        h, _, _ = _epochs(True, epochs=1)
        n0 = len(h._step_graph_cache)
        # another row total (one bag shorter): a key nobody has seen -> eager step, no new graph at first sight
        from advmil_amd import synth
        odd = [(torch.tensor([[100 + j]], dtype=torch.int), [H.T(synth.bag(H.DATA_SEED, 900 + j, m)).pin_memory(), torch.zeros(1, 1)],
                torch.tensor([[0.5, float(j % 2)]])) for j, m in enumerate((2048, 1024, 1024, 2048))]
        h.patient_id["train"] += [f"u{j}" for j in range(200)]
        h.patient_id["label_visible"] += [f"u{j}" for j in range(200)]
        cl = h._train_each_epoch(odd, "train", "wlabel")
        assert len(h._step_graph_cache) == n0 and bool(torch.isfinite(cl["y_hat"]).all())
        for g in h.optimizerG.param_groups:                       # a scheduler step: the rate is part of the key
            g["lr"] *= 0.5
        h2 = len(h._step_graph_seen)
        h._train_each_epoch([it for it in _loader_again()], "train", "wlabel")
        assert len(h._step_graph_seen) > h2
    finally:
        ops.set_gemm_mode(prev)


def _loader_again():
    from advmil_amd import synth
    i = 0
    for bl, be in zip(LENS[:4], EVENTS[:4]):
        for m, e in zip(bl, be):
            yield (torch.tensor([[i]], dtype=torch.int), [H.T(synth.bag(H.DATA_SEED, 400 + i, m)).pin_memory(), torch.zeros(1, 1)],
                   torch.tensor([[0.2 + 0.013 * i, float(e)]]))
            i += 1
