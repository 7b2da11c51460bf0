"""MyHandler.test_model (reference model_handler.py:598-643, run over the validation and the test set after every
epoch by `_run_training`, 278-285): the batched evaluation -- `batch_bags` bags through the generator and the discriminator as one
step slab, staged on the copy stream, kept in the device-resident bag cache across epochs -- against the per-bag evaluation
(`batch_bags=1`, the reference's loop shape). The reference itself pins the per-bag numbers: golden G2 (tests/test_parity_gpu.py)."""
import pytest
import torch

from advmil_amd import synth
from tests import helpers as H
from tests.test_parity_gpu import DEV, build_disc, build_generator, load_synth

pytestmark = pytest.mark.gpu


class Dataset:
    """What a torch DataLoader shows the handler: iteration + a `.dataset` object (the cache scope)."""

    def __init__(self, items, ratio_mask=None):
        self.items, self.ratio_mask = items, ratio_mask


class Loader:
    def __init__(self, dataset, order=None):
        self.dataset, self.order = dataset, order

    def __iter__(self):
        its = self.dataset.items
        return iter(its if self.order is None else [its[i] for i in self.order])


def make(kind, lens, start=40):
    items = []
    for i, n in enumerate(lens):
        ext = H.T(synth.cluster_ids(0, start + i, n)) if kind == "cluster" else torch.zeros(1, 1)
        items.append((torch.tensor([[i]], dtype=torch.int), [H.bag(start + i, max(lens))[:, :n].contiguous(), ext], H.label(start + i)))
    return items


def nets(kind, disc=("prj", "instance", "x")):
    g, d = build_generator(kind), build_disc(*disc)
    load_synth(g, f"G-{kind}:"); load_synth(d, "D-prj:" if disc[0] == "prj" else "D-cat:")
    return g, d


def same(a, b, tol=2e-6):
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and a[k].device.type == "cpu", (k, a[k].shape, b[k].shape)
        assert float((a[k].double() - b[k].double()).abs().max()) <= tol * max(1.0, float(b[k].double().abs().max())), k


@pytest.mark.parametrize("kind,disc", [("abmil", ("prj", "instance", "x")), ("patch", ("prj", "bag", "x")), ("cluster", ("cat", "bag", None))])
def test_batched_eval_equals_per_bag_eval(kind, disc):
    """Ragged bags, injected head noise (so both paths see the same draws), 7 samples per bag; batches of 3 with a remainder; one
    bag that is already a device tensor drops to the per-bag path in the middle of the epoch (order and results unchanged)."""
    from advmil_amd.model import MyHandler
    g, d = nets(kind, disc)
    lens = (256, 128, 512, 64, 192, 384, 320)
    items = make(kind, lens)
    items[4] = (items[4][0], [items[4][1][0].to(DEV), items[4][1][1]], items[4][2])
    noises = [[H.noise_tensor(f"ev:{kind}:{i}", k, 192, DEV) for k in range(8)] for i in range(len(lens))]
    per_bag = MyHandler.test_model(g, d, kind, items, times_test_sample=7, noise=noises, batch_bags=1)
    batched = MyHandler.test_model(g, d, kind, items, times_test_sample=7, noise=noises, batch_bags=3)
    same(batched, per_bag)
    assert batched["idx"].reshape(-1).tolist() == list(range(len(lens))) and batched["dist_y_hat"].shape == (len(lens), 7, 1)
    one = MyHandler.test_model(g, d, kind, items, times_test_sample=1, noise=[nz[:1] for nz in noises])
    assert set(one) == {"idx", "y", "y_hat", "f_fake"}
    same({k: one[k] for k in ("y_hat", "f_fake")}, {k: per_bag[k] for k in ("y_hat", "f_fake")})


def test_eval_bags_stay_in_hbm_between_epochs():
    """Second evaluation pass over the same dataset (another loader object, another order): no host bag is read -- the host tensors
    are poisoned in between -- and the predictions are those of the first pass, re-ordered. Slabs large enough for operand planes."""
    from advmil_amd import ingest
    from advmil_amd.model import MyHandler
    g, d = nets("abmil")
    lens = (2048, 4096, 1024, 3072, 2048)
    ds = Dataset(make("abmil", lens))
    noises = [[H.noise_tensor(f"ev2:{i}", 0, 192, DEV)] for i in range(len(lens))]
    cache = ingest.device_bag_cache(DEV)
    h0, m0 = cache.hits, cache.misses
    a = MyHandler.test_model(g, d, "abmil", Loader(ds), noise=noises, batch_bags=2)
    assert cache.misses - m0 == len(lens) and cache.hits == h0
    torch.cuda.synchronize()
    for it in ds.items:
        it[1][0].copy_(H.poison_host_bag(it[1][0]))
    order = [3, 0, 4, 1, 2]
    b = MyHandler.test_model(g, d, "abmil", Loader(ds, order), noise=[noises[i] for i in order], batch_bags=3)
    assert cache.hits - h0 == len(lens)
    assert b["idx"].reshape(-1).tolist() == order
    for k in ("y_hat", "f_fake", "y"):
        assert torch.isfinite(b[k]).all()
        assert float((b[k] - a[k][order]).abs().max()) <= 2e-6, k
    scope = ingest.dataset_scope(Loader(ds))
    del ds
    import gc
    gc.collect()
    assert not any(isinstance(k, tuple) and k[0] == scope for k in cache.entries)      # the bags went with their dataset


def test_randomly_masked_dataset_is_never_cached():
    """WSIPatch(ratio_mask=...) draws a new random instance mask per __getitem__ (dataset/PatchWSI.py:73-74): its bags are not kept."""
    from advmil_amd import ingest
    from advmil_amd.model import MyHandler
    g, d = nets("abmil")
    ds = Dataset(make("abmil", (256, 128)), ratio_mask=0.3)
    cache = ingest.device_bag_cache(DEV)
    n0, h0, m0 = len(cache.entries), cache.hits, cache.misses
    assert ingest.dataset_scope(Loader(ds)) is False
    MyHandler.test_model(g, d, "abmil", Loader(ds))
    MyHandler.test_model(g, d, "abmil", Loader(ds))
    assert (len(cache.entries), cache.hits, cache.misses) == (n0, h0, m0)


def test_cache_admission_keeps_a_resident_set_for_cohorts_larger_than_the_budget():
    """Epochs visit every bag once in a new order: with plain LRU a cohort larger than the budget gets no hits at all. The cache only
    displaces entries that have gone unused for long (a dataset no longer iterated): the hit rate over a cohort three times the budget
    is the resident third, every epoch; bags of a scope that stopped being used do give way to a new scope's."""
    import random
    from advmil_amd.ingest import BagCache
    rnd = random.Random(0)
    bag = torch.randn(1, 64, 256, device=DEV)
    c = BagCache(DEV, 20 * bag.numel() * 4)                                  # room for 20 bags
    cohort = list(range(60))
    for epoch in range(4):
        rnd.shuffle(cohort)
        h0 = c.hits
        for k in cohort:
            if c.get(("a", k)) is None:
                c.put(("a", k), bag)
        if epoch:
            assert c.hits - h0 == 20, (epoch, c.hits - h0)
    assert c.evictions == 0 and c.refused == 40 * 4 and len(c.entries) == 20
    other = list(range(100, 130))
    for epoch in range(60):                                                  # another dataset takes over: the old bags go stale
        rnd.shuffle(other)
        for k in other:
            if c.get(("b", k)) is None:
                c.put(("b", k), bag)
    mine = sum(1 for k in c.entries if k[0] == "b")
    assert mine == 20 and c.evictions == 20, (mine, c.evictions)


def test_degenerate_loaders():
    """Empty loaders, an epoch shorter than one step batch (the reference then makes no optimizer step and returns an empty collector,
    model_handler.py:321-347), a one-region bag, and an evaluation whose last slab holds a single bag."""
    from advmil_amd.config import default_cfg
    from advmil_amd.model import MyHandler
    h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=4), device=DEV)
    load_synth(h.netG, "G-abmil:"); load_synth(h.netD, "D-prj:")
    assert h._train_each_epoch([], "train") == {"y": None, "y_hat": None, "f_fake": None}
    three = make("abmil", (64, 16, 128))
    p0 = h.optimizerG.flat_param.clone()
    assert h._train_each_epoch(three, "train") == {"y": None, "y_hat": None, "f_fake": None} and torch.equal(p0, h.optimizerG.flat_param)
    assert h.pop_logs() == []
    assert MyHandler.test_model(h.netG, h.netD, "abmil", []) == {"idx": None, "y": None, "y_hat": None, "f_fake": None}
    five = make("abmil", (64, 16, 128, 16, 4112))                            # slabs of 4 + 1 bags; the last one is padded (4112 rows)
    a = MyHandler.test_model(h.netG, h.netD, "abmil", five, test_zero_noise=True, batch_bags=4)
    b = MyHandler.test_model(h.netG, h.netD, "abmil", five, test_zero_noise=True, batch_bags=1)
    same(a, b)
    cl = h._train_each_epoch(five, "train")                                    # one step of 4 bags, the fifth is dropped
    assert cl["y_hat"].shape == (4, 1) and len(h.pop_logs()) == 2 and bool(torch.isfinite(cl["f_fake"]).all())


def test_a_cache_hit_is_checked_against_the_bag_the_loader_hands_over():
    """A key that now names a DIFFERENT bag (another loader under the same scope, a dataset that changes its bags between visits)
    must not be served from the cache: the entry keeps the host bag's shape and five sampled elements and is dropped on a mismatch.
    A bare list / generator (no `.dataset`) is only cached when the configuration asks for the cache explicitly."""
    from advmil_amd.config import default_cfg
    from advmil_amd.ingest import BagCache, bag_fingerprint
    from advmil_amd.model import MyHandler
    x = torch.randn(1, 64, 256)
    c = BagCache(DEV, 1e9)
    c.put(("s", 1), x.to(DEV), bag_fingerprint(x))
    assert c.get(("s", 1), bag_fingerprint(x)) is not None and c.hits == 1
    y = x.clone()
    y[0, 0, 0] += 1.0
    assert c.get(("s", 1), bag_fingerprint(y)) is None and c.mismatches == 1 and len(c.entries) == 0 and c.bytes == 0
    c.put(("s", 1), x.to(DEV), bag_fingerprint(x))
    assert c.get(("s", 1), bag_fingerprint(x[:, :32].contiguous())) is None and c.mismatches == 2
    # list loaders: cached only on request
    lens = (256, 128)
    loader = [(torch.tensor([[i]], dtype=torch.int), [H.bag(500 + i, 256)[:, :n].contiguous(), torch.zeros(1, 1)], H.label(i))
              for i, n in enumerate(lens)]
    h = MyHandler(default_cfg(bp_every_batch=2), device=DEV)
    h._train_each_epoch(loader, "train")
    assert h._bag_caches["train"] is None
    h2 = MyHandler(default_cfg(bp_every_batch=2, bag_cache_gb=1.0), device=DEV)
    h2._train_each_epoch(loader, "train")
    h2._train_each_epoch(loader, "train")
    assert h2._bag_caches["train"] is not None and h2._bag_caches["train"].stats()["hits"] == 2
