"""Bag-parallel data parallelism: one process per GPU, one WSI (bag) per GPU at a time, replicated G and D,
one all-reduce(SUM) of each network's flat gradient arena per optimizer step over RCCL/xGMI
(torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" on CPU for tests).

The reference has no distributed path (0 collectives). World-size invariance is by construction:
 * every per-bag loss term is divided by the GLOBAL denominators the reference uses
   (#event bags / #bags / #label-visible bags of the whole step batch: model_handler.py:412, 472-478), obtained with
   one tiny all-reduce of the integer counts before the step;
 * gradients are summed, never averaged; the L1 sub-gradient is added after the reduce, identically on all ranks
   (inside the fused Adam kernel);
 * every dropout / noise draw is indexed by the row its element occupies in the SINGLE-PROCESS step slab (bags in global order
   i = local_index * W + rank): the handler's step plan all-gathers the bag lengths and hands the kernels per-row maps
   (ops.DeviceRng.rows keyed by layout kind, selected by the call site's tag -- ops.SITE_LAYOUTS --; the rng_row arguments of the
   C ABI), so a W-rank step draws exactly the masks of the 1-rank step; a site without a registered layout raises at world > 1;
 * the epoch collector (y, y_hat, f_fake) is all-gathered back into global bag order; logged losses are all-reduced.
This module is compute-agnostic (it only sees flat tensors), so the gloo tests drive it on CPU.
"""
import os

import torch
import torch.distributed as dist


class BagParallel:
    def __init__(self, group=None, force=False):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1
        # force: issue the step's collectives also on a ONE-rank communicator (a test aid: this image has one GPU per box, so the
        # captured-collective path of graphed.GraphedStep can only be exercised through a one-rank RCCL group)
        self.force = bool(force) and self.enabled

    # ---- partition: bag i of the global step batch -> rank i mod W ("one WSI per GPU")
    def owns(self, global_index: int) -> bool:
        return global_index % self.world == self.rank

    def shard(self, items):
        return [it for i, it in enumerate(items) if self.owns(i)]

    def global_index(self, local_index: int) -> int:
        return local_index * self.world + self.rank

    def local_step_bags(self, bp_every_batch: int) -> int:
        """cfg['bp_every_batch'] is the GLOBAL step batch (config/cfg_nlst.yaml:71: 16 bags per optimizer step), whatever the world
        size: every rank steps after bp_every_batch / W of its own bags."""
        if bp_every_batch % self.world != 0:
            raise ValueError(f"bp_every_batch={bp_every_batch} is the global step batch and must be a multiple of the world size "
                             f"{self.world} (bag i of a step batch runs on rank i mod W)")
        return bp_every_batch // self.world

    def shard_epoch(self, items, bp_every_batch: int):
        """This rank's bags of an epoch: the reference steps every bp_every_batch bags and never back-propagates a trailing partial
        batch (model_handler.py:321-345), so only the first floor(len / bp) * bp bags count; of those, bag i runs on rank i mod W.
        Every rank gets the same number of bags and of optimizer steps (a rank with one bag more would hang the others in the
        step's collectives)."""
        self.local_step_bags(bp_every_batch)
        usable = len(items) // bp_every_batch * bp_every_batch
        return [it for i, it in enumerate(items[:usable]) if self.owns(i)]

    def check_equal(self, value: int, what: str, device="cpu"):
        """Raise on every rank if `value` differs between ranks (instead of hanging in a later collective)."""
        if not (self.enabled and self.world > 1):
            return
        vals = [v[0] for v in self.allgather_ints([int(value)], device)]
        if len(set(vals)) != 1:
            raise RuntimeError(f"bag-parallel: {what} differs between ranks ({vals}); shard the epoch with "
                               "BagParallel.shard_epoch so that every rank runs the same number of optimizer steps")

    # ---- exchange
    def allreduce_(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place SUM of a flat gradient arena across ranks (one bucket per network)."""
        if self.enabled and (self.world > 1 or self.force):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def allreduce_async(self, *tensors):
        """Start SUM all-reduces and return their work handles: with RCCL they run on the communicator's own stream, `wait()`
        then only makes the current compute stream wait (no host block), so kernels enqueued in between overlap the exchange."""
        if not (self.enabled and (self.world > 1 or self.force)):
            return []
        return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for t in tensors]

    def global_counts(self, counts, device="cpu"):
        """Sum small integer counters ([n_real, n_fake, n_visible]) across ranks -> python ints."""
        if not (self.enabled and self.world > 1):
            return [int(c) for c in counts]
        t = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(round(v)) for v in t.tolist()]

    def allgather_ints(self, vals, device="cpu"):
        """Every rank's list of small integers (equal lengths) -> [W][len] python lists (one tiny all-gather)."""
        if not (self.enabled and self.world > 1):
            return [[int(v) for v in vals]]
        t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.group)
        return [o.tolist() for o in outs]

    def allgather_cat(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenate equally-shaped per-rank tensors in global bag order (for the epoch collector)."""
        if not (self.enabled and self.world > 1):
            return t
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t.contiguous(), group=self.group)
        return torch.stack(outs, dim=1).reshape(-1, *t.shape[1:])   # interleave: local j of rank r -> j*W + r

    def broadcast_(self, flat: torch.Tensor, src=0):
        if self.enabled and self.world > 1:
            dist.broadcast(flat, src=src, group=self.group)
        return flat


def rng_row_maps(all_lens, W, r, cluster=False, pads=None):
    """Host side of the world-size-invariant randomness. `all_lens[q][j]` = patch rows of rank q's j-th bag of the step; this
    rank's bag j is bag j*W + r of the global step batch, whose single-process slab stacks the bags in global order.
    Returns ({layout kind: int64 array local row -> single-process row}, per-bag region-row offsets for the attention kernels).
    Kinds (the layouts a slab-level tensor of the step can have; every dropout / noise call site names the kinds it can be fed,
    ops.SITE_LAYOUTS, so two layouts that happen to have the same row count never get confused):
      patch [sum N], region [sum N/16], region2 [2 sum N/16] (the D update's stacked fake|real region rows), bag [n],
      bag2 [2n] (the stacked tail rows), cluster [8n] (DeepAttMISL's cluster rows).
    `pads[q]`: zero rows rank q appends to its slab (ingest.SlabStager.pad_rows: a dummy bag behind the real ones). Its rows draw at
    indices BEHIND every real row of the single-process slab, so the real rows' draws do not depend on the pad either (W = 1 with a
    pad uses these maps too: a padded step draws exactly what the unpadded step draws). The dummy bag is dropped before any
    bag-level site, so `bag` / `bag2` stay [n] / [2n]; `cluster` and the attention offsets get one more bag."""
    import numpy as np
    lens = list(all_lens[r])
    n = len(lens)
    G = n * W
    glens = [all_lens[gi % W][gi // W] for gi in range(G)]
    goff = np.concatenate([[0], np.cumsum(glens)]).astype(np.int64)
    gi = [j * W + r for j in range(n)]
    patch = np.concatenate([goff[gi[j]] + np.arange(lens[j], dtype=np.int64) for j in range(n)])
    region = np.concatenate([goff[gi[j]] // 16 + np.arange(lens[j] // 16, dtype=np.int64) for j in range(n)])
    bags = np.asarray(gi, dtype=np.int64)
    SLg = int(goff[-1]) // 16
    pad = 0 if pads is None else int(pads[r])
    if pad:
        before, ptot = int(sum(pads[:r])), int(sum(pads))
        p16 = (pad + 15) // 16
        b16, t16 = (before + 15) // 16 + r, (ptot + 15) // 16 + W           # (disjoint per rank whatever the pads' remainders)
        patch = np.concatenate([patch, int(goff[-1]) + before + np.arange(pad, dtype=np.int64)])
        padA = 2 * SLg + b16 + np.arange(p16, dtype=np.int64)
        region2 = np.concatenate([region, padA, SLg + region, padA + t16])
        region = np.concatenate([region, padA])
    else:
        region2 = np.concatenate([region, SLg + region])
    maps = {"patch": patch, "region": region, "region2": region2, "bag": bags, "bag2": np.concatenate([bags, G + bags])}
    if cluster:
        cl = [8 * g_ + np.arange(8, dtype=np.int64) for g_ in gi]
        if pad:
            cl.append(8 * (G + r) + np.arange(8, dtype=np.int64))
        maps["cluster"] = np.concatenate(cl)
    seg_lens = lens + [pad] if pad else lens
    loc16 = np.concatenate([[0], np.cumsum([v // 16 for v in seg_lens])])[:-1]
    first16 = [goff[gi[j]] // 16 for j in range(n)] + ([2 * SLg + b16] if pad else [])
    off16 = np.asarray([first16[j] - loc16[j] for j in range(len(seg_lens))], dtype=np.int64)
    return maps, off16


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("ADVMIL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            kw["device_id"] = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local
