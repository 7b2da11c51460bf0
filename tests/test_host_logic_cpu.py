"""CPU: host-side logic of the path that needs no GPU -- config slicing helpers, synthetic-data determinism,
the counter RNG restatement, module construction / state_dict surface, split heuristic."""
import numpy as np
import torch

from advmil_amd import synth
from advmil_amd.config import default_cfg
from advmil_amd.utils.func import agg_tensor, collect_tensor, sparse_key, sparse_str
from tests import helpers as H


def test_sparse_helpers_match_reference_semantics():
    cfg = default_cfg()
    assert sparse_str(cfg["bcb_dims"]) == [1024, 384, 384]
    assert sparse_str(0.5) == [0.5]
    k = sparse_key(cfg, prefixes="disc_netx")
    assert k == {"in_dim": 1024, "out_dim": 128, "ksize": 1, "backbone": "avgpool", "dropout": 0.25}
    assert sparse_key(cfg, "loss_recon") == {"norm": "l1", "alpha": 0.0, "gamma": 0.0}
    assert sparse_key(cfg, "gen_noi") == {"noise": "0-1", "noise_dist": "uniform", "hops": 1}


def test_collectors():
    c = {"real": None, "fake": None}
    c = collect_tensor(c, None, torch.ones(2))
    c = collect_tensor(c, torch.zeros(1), torch.ones(1))
    assert c["real"].shape == (1,) and c["fake"].shape == (3,)
    a = agg_tensor({"y": None}, {"y": torch.ones(2, 2)})
    a = agg_tensor(a, {"y": torch.ones(1, 2)})
    assert a["y"].shape == (3, 2)


def test_synth_is_deterministic_and_well_distributed():
    a = synth.bag(0, 3, 512)
    b = synth.bag(0, 3, 512)
    assert np.array_equal(a, b) and a.shape == (1, 512, 1024) and a.dtype == np.float32
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1.0) < 0.01
    assert not np.array_equal(a, synth.bag(0, 4, 512))
    y = synth.label(0, 5)
    assert y.shape == (1, 2) and y[0, 1] == 1.0 and 0.05 <= y[0, 0] <= 0.95
    u = synth.kernel_uniform(7, 3, 100000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    keep = synth.dropout_keep(7, 3, 100000, 0.25)
    assert abs(keep.mean() - 0.75) < 0.01
    # known-answer: splitmix64(0) (the published first output of the generator seeded with 0)
    assert int(synth.splitmix64(np.uint64(0))) == 0xE220A8397B1DCDAF


def test_grid_graph_shape():
    e = synth.grid_knn_graph(100, 8)
    assert e.shape == (2, 800) and e.min() >= 0 and e.max() < 100
    assert np.array_equal(e[0], np.repeat(np.arange(100), 8))


def test_state_dict_surface_on_cpu():
    from types import SimpleNamespace
    from advmil_amd.model import Generator, PrjDiscriminator, Discriminator, load_backbone
    for kind in ("abmil", "patch", "cluster"):
        g = Generator(384, 1, load_backbone(kind, [1024, 384, 384]), SimpleNamespace(noise=[0, 1], hops=1, noise_dist=None),
                      False, 0.6, "sigmoid")
        want = H.shapes_generator(kind)
        got = {k: tuple(v.shape) for k, v in g.state_dict().items()}
        assert got == want, kind
    ax = SimpleNamespace(in_dim=1024, out_dim=128, ksize=1, backbone="avgpool", dropout=0.25)
    ay = SimpleNamespace(in_dim=1, hid_dims=[64, 128], norm=False, dropout=0.0)
    assert {k: tuple(v.shape) for k, v in PrjDiscriminator(ax, ay, "x", "instance").state_dict().items()} == H.shapes_disc("prj", "x")
    assert {k: tuple(v.shape) for k, v in Discriminator(ax, ay).state_dict().items()} == H.shapes_disc("cat", None)
    # parameter counts probed from the reference
    n = lambda sh: sum(int(np.prod(s)) for s in sh.values())
    assert n(H.shapes_generator("abmil")) == 911810 and n(H.shapes_generator("patch")) == 1653314
    assert n(H.shapes_disc("prj", "x")) == 206338


def test_auto_splits_heuristic():
    from advmil_amd.ops import auto_splits
    assert auto_splits(8192, 384, 1024) == 1            # 192 tiles: enough parallelism
    assert auto_splits(384, 1024, 8192) > 1             # dW: 24 tiles, the bag length is K
    assert auto_splits(8, 384, 64) == 1


def test_host_copy_rows_and_effective_cpus(monkeypatch):
    """ingest.host_copy_rows: the pageable -> pinned copy of a loader tensor by a small thread pool (chunks of rows; any row count,
    also fewer rows than threads, non-contiguous sources and a different dtype take torch's copy_) copies exactly; effective_cpus is
    the affinity mask capped by the cgroup quota and at least 1."""
    import os
    from advmil_amd import ingest
    assert 1 <= ingest.effective_cpus() <= (os.cpu_count() or 1)
    g = torch.Generator().manual_seed(0)
    for nt in ("8", "3", "1", "0"):
        monkeypatch.setenv("ADVMIL_INGEST_THREADS", nt)
        for rows in (1, 5, 33, 257, 1000):
            src = torch.randn(rows, 96, generator=g)
            dst = torch.full((rows, 96), float("nan"))
            ingest.host_copy_rows(dst, src)
            assert torch.equal(dst, src), (nt, rows)
        src = torch.randn(64, 192, generator=g)[:, ::2]                       # non-contiguous source
        dst = torch.empty(64, 96)
        ingest.host_copy_rows(dst, src)
        assert torch.equal(dst, src)
        src = torch.randn(64, 96, generator=g).double()                        # another dtype: converted by copy_
        dst = torch.empty(64, 96)
        ingest.host_copy_rows(dst, src)
        assert torch.equal(dst, src.float())


def test_dataset_scope_follows_the_dataset_object():
    """ingest.dataset_scope: the bag cache's scope of a loader is its DATASET object (two loaders over one dataset share it, two
    datasets never do, even at a recycled address), None without one, False for a dataset that re-draws a random instance mask per
    visit (WSIPatch.ratio_mask, dataset/PatchWSI.py:73-74); the token dies with the object."""
    import gc
    from types import SimpleNamespace
    from advmil_amd import ingest

    class DS:
        def __init__(self, ratio_mask=None):
            self.ratio_mask = ratio_mask

    a, b = DS(), DS()
    sa = ingest.dataset_scope(SimpleNamespace(dataset=a))
    assert sa == ingest.dataset_scope(SimpleNamespace(dataset=a)) and sa != ingest.dataset_scope(SimpleNamespace(dataset=b))
    assert ingest.dataset_scope([1, 2, 3]) is None and ingest.dataset_scope(SimpleNamespace(dataset=DS(0.3))) is False
    assert ingest.dataset_scope(SimpleNamespace(dataset=DS(0.0))) not in (None, False)          # ratio_mask 0: nothing is masked
    n0 = len(ingest._SCOPE_TOKENS)
    ida = id(a)
    del a
    gc.collect()
    assert ida not in ingest._SCOPE_TOKENS and len(ingest._SCOPE_TOKENS) < n0 + 1
    seen = {sa}
    for _ in range(50):                                                         # fresh objects (often at the address just freed): fresh tokens
        s = ingest.dataset_scope(SimpleNamespace(dataset=DS()))
        assert s not in seen
        seen.add(s)
