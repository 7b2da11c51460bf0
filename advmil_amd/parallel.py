"""Bag-parallel data parallelism: one process per GPU, one WSI (bag) per GPU at a time, replicated G and D,
one all-reduce(SUM) of each network's flat gradient arena per optimizer step over RCCL/xGMI
(torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" on CPU for tests).

The reference has no distributed path (0 collectives). World-size invariance is by construction:
 * every per-bag loss term is divided by the GLOBAL denominators the reference uses
   (#event bags / #bags / #label-visible bags of the whole step batch: model_handler.py:412, 472-478), obtained with
   one tiny all-reduce of the integer counts before the step;
 * gradients are summed, never averaged; the L1 sub-gradient is added after the reduce, identically on all ranks
   (inside the fused Adam kernel);
 * per-bag dropout/noise streams derive from (seed, global bag index), not from the rank.
This module is compute-agnostic (it only sees flat tensors), so the gloo tests drive it on CPU.
"""
import os

import torch
import torch.distributed as dist


class BagParallel:
    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1

    # ---- partition: bag i of the global step batch -> rank i mod W ("one WSI per GPU")
    def owns(self, global_index: int) -> bool:
        return global_index % self.world == self.rank

    def shard(self, items):
        return [it for i, it in enumerate(items) if self.owns(i)]

    def global_index(self, local_index: int) -> int:
        return local_index * self.world + self.rank

    # ---- exchange
    def allreduce_(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place SUM of a flat gradient arena across ranks (one bucket per network)."""
        if self.enabled and self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def global_counts(self, counts, device="cpu"):
        """Sum small integer counters ([n_real, n_fake, n_visible]) across ranks -> python ints."""
        if not (self.enabled and self.world > 1):
            return [int(c) for c in counts]
        t = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(round(v)) for v in t.tolist()]

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
