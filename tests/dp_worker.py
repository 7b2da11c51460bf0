"""One rank of the bag-parallel product-handler test (tests/test_parallel_gpu.py): MyHandler on cuda:0 with dropout ON, this rank's
shard of the global step batches (bag i on rank i mod W), two optimizer steps through _train_each_epoch; rank 0 saves what the
single-process run is compared with. usage: python -m tests.dp_worker RANK WORLD PORT OUT KIND"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LENS = (256, 128, 64, 192, 320, 96, 160, 224)       # 2 steps x 4 bags (global), ragged, multiples of 16
# `collide`: per rank 2 bags of 32 patches per step -> 2n = 4 stacked tail rows AND sum N / 16 = 4 region rows: two different
# layouts with the same row count (the row maps are keyed by call site, not by row count)
LENS_COLLIDE = (32, 32, 32, 32, 32, 32, 32, 32)
# `pad`: slabs large enough for the zero-row pad of the staging slab (>= 4096 rows, not a multiple of 256) -- with two ranks each
# pads its own half of the step batch (different pads per rank and per step), the single process pads the whole batch
LENS_PAD = (2064, 2208, 2576, 2128, 2320, 2096, 2704, 2240)


# `bp16`: the shipped step batch (cfg_nlst.yaml:71: bp_every_batch = 16) -- 2 steps x 16 ragged bags, for the W = 4 / W = 8 cases (4 / 2
# bags per rank and step: the split SURVEY section 8e prescribes for the 8-GPU node)
LENS_BP16 = (256, 128, 64, 192, 320, 96, 160, 224, 48, 272, 112, 208, 80, 304, 144, 176,
             240, 32, 288, 128, 64, 336, 96, 192, 160, 16, 224, 256, 112, 80, 208, 144)


# `big`: slab-sized bags (2 steps x 4 bags of ~8192 patches): with two ranks each rank's step slab is 16 384 rows -- the size class where the
# two-layer launch, the planes-only first layer, the dropout of planes with the scorer's keep bits and the fused training gate score all run
# (round 6), each with this rank's dropout row map
LENS_BIG = (8192, 8176, 8208, 8192, 8160, 8224, 8192, 8192)


def build_loader(kind, idxs, lens=LENS):
    from types import SimpleNamespace
    from advmil_amd import synth
    from tests import helpers as H
    loader = []
    for i in idxs:
        x = H.bag(300 + i, max(512, max(lens)))[:, :lens[i]].contiguous() if max(lens) <= 4096 else H.bag(300 + i, lens[i])
        if kind == "graph":                  # PatchGCN: device-resident graph objects (x [N, C], edge_index [2, 8N]), as the bench feeds them
            x = x.to("cuda:0")
            ext = SimpleNamespace(x=x[0], edge_index=H.T(synth.grid_knn_graph(lens[i], 8), "cuda:0"))
        else:
            ext = H.T(synth.cluster_ids(0, 300 + i, lens[i])) if kind == "cluster" else torch.zeros(1, 1)
        loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext], H.label(300 + i)))
    return loader


def run(kind, world, rank, dp=None, device="cuda:0"):
    from advmil_amd import synth
    from advmil_amd.config import default_cfg
    from advmil_amd.model import MyHandler
    from tests import helpers as H
    lens, bp = LENS, 4
    if kind.endswith("-bp16"):
        kind, lens, bp = kind[:-len("-bp16")], LENS_BP16, 16
    elif kind.endswith("-big"):
        kind, lens = kind[:-len("-big")], LENS_BIG
    elif kind.endswith("-collide"):
        kind, lens = kind[:-len("-collide")], LENS_COLLIDE
    elif kind.endswith("-pad"):
        kind, lens = kind[:-len("-pad")], LENS_PAD
    elif kind.endswith("-env"):                      # tools/probe/dp_fuzz.py: eight bag lengths from the environment
        kind, lens = kind[:-len("-env")], tuple(int(v) for v in os.environ["DP_LENS"].split(","))
    cfg = default_cfg(bcb_mode=kind, bp_every_batch=bp)        # the GLOBAL step batch: every rank steps after bp / world of its bags
    if lens is LENS_BIG:
        cfg["gemm_mode"] = "bf16x3"          # the product's default arithmetic: the operand-plane paths only exist in it
    if kind == "graph":
        cfg.update(bcb_dims="1024-128-128", gen_dims="128-1")
    from advmil_amd import ops as _ops
    mode0 = _ops.get_gemm_mode()             # (a handler sets the library's arithmetic mode from its cfg: restored below, so that the single-process
    h = MyHandler(cfg, device=device, parallel=dp)       # run inside the pytest process does not leave bf16x3 behind for the files after this one)
    for net, prefix in ((h.netG, f"G-{kind}:"), (h.netD, "D-prj:")):
        sd = {k: H.T(synth.param(H.PARAM_SEED, prefix + k, tuple(v.shape))) for k, v in net.state_dict().items()}
        net.load_state_dict(sd, strict=True)
    h.rng.reset(4321)
    from advmil_amd.parallel import BagParallel
    idxs = (dp or BagParallel()).shard_epoch(list(range(len(lens))), bp)    # bag i of a global step batch -> rank i mod W
    from advmil_amd import ops
    score_rows, real_score = [], ops.gate_score
    ops.gate_score = lambda *a, **k: (score_rows.append(int(a[3])), real_score(*a, **k))[1]      # (which slabs still take the score PASS)
    try:
        cl = h._train_each_epoch(build_loader(kind, idxs, lens), "train", "wlabel")
    finally:
        ops.gate_score = real_score
        ops.set_gemm_mode(mode0)
    logs = h.pop_logs()
    return {"cl": cl, "logs": logs, "gate_score_rows": score_rows, "G": {k: v.detach().cpu() for k, v in h.netG.state_dict().items()},
            "D": {k: v.detach().cpu() for k, v in h.netD.state_dict().items()}}


def main():
    rank, world, port, out, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    per_rank = os.environ.get("ADVMIL_DP_DEVICE_PER_RANK") == "1"       # one GPU per rank + RCCL (a multi-GPU node), else ranks share cuda:0 over gloo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank) if per_rank else "0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from advmil_amd import parallel
    # two ranks on ONE GPU: RCCL refuses that, the exchange layer is backend agnostic
    parallel.init_from_env(backend=os.environ.get("ADVMIL_DIST_BACKEND", "gloo") if per_rank else "gloo")
    res = run(kind, world, rank, parallel.BagParallel(), device=f"cuda:{rank}" if per_rank else "cuda:0")
    if rank == 0:
        torch.save(res, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
