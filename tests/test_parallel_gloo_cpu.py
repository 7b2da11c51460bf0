"""CPU, world_size 2 over gloo: the bag-parallel exchange layer (advmil_amd/parallel.py) is world-size invariant.
Compute here is the oracle (tests may use it); the exchange, partition, global-denominator and gather logic is the
product's. The 2-rank run must reproduce the 1-rank oracle step: same losses, same summed gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from advmil_amd import parallel
    from oracle import advmil_oracle as O
    from tests import helpers as H
    parallel.init_from_env(backend="gloo")
    dp = parallel.BagParallel()
    assert dp.world == world and dp.rank == rank
    kind, nb, N = "abmil", 4, 64
    PG = H.synth_params(H.shapes_generator(kind), f"G-{kind}:")
    PD = H.synth_params(H.shapes_disc(), "D-prj:")
    bags = [(H.bag(i, N), None, H.label(i)) for i in range(nb)]
    nd = [[H.noise_tensor("dp_d", i, 192)] for i in range(nb)]
    ng = [[H.noise_tensor("dp_g", i, 192)] for i in range(nb)]
    cfg = O.StepConfig(kind=kind)
    mine = [i for i in range(nb) if dp.owns(i)]
    assert dp.shard(list(range(nb))) == mine and [dp.global_index(j) for j in range(len(mine))] == mine
    lb, lnd, lng = [bags[i] for i in mine], [nd[i] for i in mine], [ng[i] for i in mine]
    n_real_l = sum(int(b[2][0, 1] == 1) for b in lb)
    n_real, n_fake = dp.global_counts([n_real_l, len(lb)])
    assert (n_real, n_fake) == (2, 4)
    # ---- D phase on the shard with global denominators, then ONE all-reduce of the flat grad arena
    logs_d, gD, preds, fakes = O.update_disc(cfg, PG, PD, lb, lnd, n_real_global=n_real, n_fake_global=n_fake)
    keys = sorted(PD)
    flat = torch.cat([gD.get(k, torch.zeros_like(PD[k])).reshape(-1) for k in keys])
    dp.allreduce_(flat)
    loss = torch.tensor([logs_d["Loss_D"]], dtype=torch.float64)
    dist.all_reduce(loss)
    # ---- G phase
    logs_g, gG, _ = O.update_gen(cfg, PG, PD, lb, lng, n_global=n_fake)
    keysg = sorted(PG)
    flatg = torch.cat([gG[k].reshape(-1) for k in keysg])
    dp.allreduce_(flatg)
    yh = dp.allgather_cat(torch.cat(preds))            # epoch collector in global bag order
    lens_all = dp.allgather_ints([100 + rank, 7 * (rank + 1)])
    assert lens_all == [[100, 7], [101, 14]]
    if rank == 0:
        q.put({"flat_d": flat.numpy().copy(), "flat_g": flatg.numpy().copy(), "loss_d": float(loss), "y_hat": yh.numpy().copy()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_rank():
    sys.path.insert(0, ROOT)
    from oracle import advmil_oracle as O
    from tests import helpers as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    got = {k: (torch.from_numpy(v) if not isinstance(v, float) else v) for k, v in got.items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    kind, nb, N = "abmil", 4, 64
    PG = H.synth_params(H.shapes_generator(kind), f"G-{kind}:")
    PD = H.synth_params(H.shapes_disc(), "D-prj:")
    bags = [(H.bag(i, N), None, H.label(i)) for i in range(nb)]
    nd = [[H.noise_tensor("dp_d", i, 192)] for i in range(nb)]
    ng = [[H.noise_tensor("dp_g", i, 192)] for i in range(nb)]
    cfg = O.StepConfig(kind=kind)
    logs_d, gD, preds, _ = O.update_disc(cfg, PG, PD, bags, nd)
    flat = torch.cat([gD.get(k, torch.zeros_like(PD[k])).reshape(-1) for k in sorted(PD)])
    cfg_nol1 = O.StepConfig(kind=kind, l1_coef=0.0)     # the shard backward leaves L1 to the optimizer kernel
    _, gG, _ = O.update_gen(cfg_nol1, PG, PD, bags, ng)
    flatg = torch.cat([gG[k].reshape(-1) for k in sorted(PG)])
    assert abs(got["loss_d"] - logs_d["Loss_D"]) < 1e-6
    assert float((got["flat_d"] - flat).abs().max()) < 1e-6 * (1 + float(flat.abs().max()))
    assert float((got["flat_g"] - flatg).abs().max()) < 1e-6 * (1 + float(flatg.abs().max()))
    assert float((got["y_hat"].reshape(-1) - torch.cat(preds).reshape(-1)).abs().max()) < 1e-6


def test_rng_row_maps_cover_the_single_process_slab():
    """parallel.rng_row_maps: the ranks' local rows map onto a partition of the single-process slab's rows, bag j of rank r being
    global bag j*W + r; the stacked (fake | real) layouts map half by half; the attention kernels' per-bag offsets agree."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from advmil_amd.parallel import rng_row_maps
    W = 2
    all_lens = [[256, 64, 128], [32, 512, 96]]          # rank 0's and rank 1's bags of one step (global order interleaves them)
    glob = [all_lens[g % W][g // W] for g in range(6)]   # 256, 32, 64, 512, 128, 96
    goff = np.concatenate([[0], np.cumsum(glob)])
    seen_patch, seen_region, seen_bag = [], [], []
    for r in range(W):
        maps, off16 = rng_row_maps(all_lens, W, r, cluster=True)
        lens = all_lens[r]
        n, SN, SL = len(lens), sum(lens), sum(lens) // 16
        assert {k: len(v) for k, v in maps.items()} == {"patch": SN, "region": SL, "region2": 2 * SL, "bag": n, "bag2": 2 * n, "cluster": 8 * n}
        o = 0
        for j, L in enumerate(lens):                      # local bag j = global bag j*W + r, rows in order
            g = j * W + r
            assert np.array_equal(maps["patch"][o:o + L], goff[g] + np.arange(L))
            assert off16[j] == goff[g] // 16 - o // 16
            o += L
        assert np.array_equal(maps["region2"][:SL], maps["region"]) and np.array_equal(maps["region2"][SL:], maps["region"] + goff[-1] // 16)
        assert np.array_equal(maps["bag"], [j * W + r for j in range(n)])
        assert np.array_equal(maps["bag2"][n:], maps["bag"] + 6)
        assert np.array_equal(maps["cluster"].reshape(n, 8), 8 * maps["bag"][:, None] + np.arange(8)[None, :])
        seen_patch.append(maps["patch"]); seen_region.append(maps["region"]); seen_bag.append(maps["bag"])
    assert np.array_equal(np.sort(np.concatenate(seen_patch)), np.arange(goff[-1]))
    assert np.array_equal(np.sort(np.concatenate(seen_region)), np.arange(goff[-1] // 16))
    assert np.array_equal(np.sort(np.concatenate(seen_bag)), np.arange(6))
    # 2 regions per bag: the [2n] stacked-tail rows and the region rows have the same COUNT; the maps are per layout, so both exist
    maps, _ = rng_row_maps([[32, 32], [32, 32]], 2, 0)
    assert len(maps["bag2"]) == len(maps["region"]) == 4 and not np.array_equal(maps["bag2"], maps["region"])


def test_row_map_is_selected_by_call_site_not_by_row_count():
    """ops.DeviceRng.row_map: the site's tag names the layouts it can be fed; equal row counts of two different layouts do not
    collide, and a site without a registered layout (or fed an unexpected row count) raises at world > 1 instead of silently
    drawing local-row masks."""
    sys.path.insert(0, ROOT)
    from advmil_amd import ops
    from advmil_amd.parallel import rng_row_maps
    maps, _ = rng_row_maps([[32, 32], [32, 32]], 2, 1)
    rng = ops.DeviceRng("cpu")
    assert rng.row_map(4, "dx_fc2.2") is None                                  # single process: identity
    rng.rows = {k: torch.from_numpy(v) for k, v in maps.items()}
    assert torch.equal(rng.row_map(4, "dx_fc2.2"), rng.rows["bag2"])           # 4 stacked tail rows
    assert torch.equal(rng.row_map(4, "gapool_att_a"), rng.rows["region"])     # 4 region rows: same count, other layout
    assert torch.equal(rng.row_map(8, "dx_fc1"), rng.rows["region2"])
    assert torch.equal(rng.row_map(64, "abmil_fc"), rng.rows["patch"])
    assert torch.equal(rng.row_map(2, "gen_mlp0.2"), rng.rows["bag"])
    with pytest.raises(RuntimeError, match="no registered row layout"):
        rng.row_map(4, "some_new_dropout_site")
    with pytest.raises(RuntimeError, match="none of its layouts"):
        rng.row_map(5, "abmil_fc")
    # a single process whose slab carries a zero-row pad maps only the stacked layout; the others are declared identities (no lookup in
    # the kernels) but still checked for their row count
    rng.rows = {"patch": ops.IdentityRows(80), "region": ops.IdentityRows(5), "bag": ops.IdentityRows(2), "bag2": ops.IdentityRows(4),
                "region2": torch.arange(10)}
    assert rng.row_map(80, "abmil_fc") is None and rng.row_map(5, "gapool_att_a") is None and rng.row_map(4, "dx_fc2.2") is None
    assert torch.equal(rng.row_map(10, "dx_fc1"), torch.arange(10))
    with pytest.raises(RuntimeError, match="none of its layouts"):
        rng.row_map(64, "abmil_fc")


def test_shard_epoch_gives_every_rank_the_same_number_of_steps():
    """BagParallel.shard_epoch: the trailing partial step batch is dropped (the reference never back-propagates it,
    model_handler.py:321-345), bag i of a step batch goes to rank i mod W, cfg['bp_every_batch'] stays the GLOBAL step batch."""
    sys.path.insert(0, ROOT)
    from advmil_amd.parallel import BagParallel
    items = list(range(37))
    shards = []
    for r in range(4):
        dp = BagParallel()
        dp.world, dp.rank = 4, r
        assert dp.local_step_bags(16) == 4
        shards.append(dp.shard_epoch(items, 16))
        with pytest.raises(ValueError):
            dp.local_step_bags(6)
    assert all(len(s) == 8 for s in shards)                  # 37 bags -> 2 global steps of 16 -> 8 bags per rank
    assert sorted(sum(shards, [])) == list(range(32))
    assert shards[1][:4] == [1, 5, 9, 13] and shards[1][4:] == [17, 21, 25, 29]


def test_row_maps_with_slab_pads_leave_the_real_rows_draws_unchanged():
    """A rank may append zero rows (a dummy bag) to its slab so that the slab kernels see whole tiles (ingest.SlabStager.pad_rows):
    the real rows keep the single-process indices they have without the pad, the pad rows get indices behind every real row,
    disjoint between ranks and between the two copies of the stacked region layout; W = 1 with a pad is the identity on real rows."""
    import numpy as np
    from advmil_amd.parallel import rng_row_maps
    all_lens = [[64, 32], [48, 96]]
    pads = [16, 160]
    total = sum(sum(v) for v in all_lens)
    seen_patch, seen_r2 = [], []
    for r in range(2):
        base, off0 = rng_row_maps(all_lens, 2, r, cluster=True)
        padded, off1 = rng_row_maps(all_lens, 2, r, cluster=True, pads=pads)
        nreal, lreal = sum(all_lens[r]), sum(all_lens[r]) // 16
        assert np.array_equal(padded["patch"][:nreal], base["patch"]) and padded["patch"].shape[0] == nreal + pads[r]
        assert padded["patch"][nreal:].min() >= total
        assert np.array_equal(padded["region"][:lreal], base["region"]) and padded["region"].shape[0] == lreal + pads[r] // 16
        p16 = pads[r] // 16
        r2 = padded["region2"]
        assert r2.shape[0] == 2 * (lreal + p16)
        assert np.array_equal(np.concatenate([r2[:lreal], r2[lreal + p16:2 * lreal + p16]]), base["region2"])
        assert np.array_equal(padded["bag"], base["bag"]) and np.array_equal(padded["bag2"], base["bag2"])
        assert np.array_equal(padded["cluster"][:16], base["cluster"]) and padded["cluster"].shape[0] == 24
        assert np.array_equal(off1[:2], off0) and off1.shape[0] == 3
        seen_patch.append(padded["patch"][nreal:]); seen_r2.append(np.concatenate([r2[lreal:lreal + p16], r2[2 * lreal + p16:]]))
        assert padded["region2"][lreal:lreal + p16].min() >= 2 * (total // 16)
    assert len(np.unique(np.concatenate(seen_patch))) == sum(pads) and len(np.unique(np.concatenate(seen_r2))) == 2 * sum(pads) // 16
    one, _ = rng_row_maps([[64, 32]], 1, 0, pads=[160])
    assert np.array_equal(one["patch"][:96], np.arange(96)) and np.array_equal(one["region2"][6 + 10:6 + 10 + 6], 6 + np.arange(6))
