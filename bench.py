#!/usr/bin/env python3
"""bench.py -- WSI bags/sec through the full AdvMIL G+D training step on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1 under torch.distributed.run (RANK / WORLD_SIZE in the environment): this process is one rank.
      N > 1 WITHOUT a launcher environment: bench.py starts the N ranks itself (a child `python -m torch.distributed.run
      --nproc-per-node N bench.py ...`, spawned before this process touches the GPU), relays rank 0's JSON line and the
      exit code. It refuses (exit 2) when the box has fewer than N GPUs, unless ADVMIL_DIST_BACKEND=gloo asks for ranks
      that share devices (functional testing).

One "step" = one optimizer step of the hot path (`_update_disc` + `gen_updates=1` x `_update_gen`,
reference model/model_handler.py:321-345) over `--bags` bags per GPU (default bp_every_batch = 16,
config/cfg_nlst.yaml:71). Workload = BASELINE.json configs[1]: ABMIL generator + RLIP projection
discriminator on synthetic 8192-patch x 1024 bags, resident in HBM before the timed region
(>= 64 distinct bags per GPU = 2.1 GB >> the 256 MB Infinity Cache). Multi-GPU: bag-parallel, every rank
runs its own bags, one RCCL all-reduce of each network's flat gradient arena per step. `value` is the WEAK-scaling figure
(global step batch = bags x N); the line also carries `strong_scaling` (N > 1): the reference's own step batch of 16 bags
(cfg_nlst.yaml:71) split over the ranks, 16 / N bags per rank per step.

Prints ONE JSON line (rank 0). Extra objects:
  roofline     : the dominant kernel of the step. ABMIL / DeepAttMISL / PatchGCN: the contraction (kernel, shape) that owns the
                 most device time in an instrumented pass (HIP events around every launch on the launch stream), then timed as 50
                 back-to-back launches; algorithmic 2MNK flops / duration against the MFMA roof of the arithmetic mode.
                 ESAT (--mode patch): the fused attention core (advmil_mha_fwd + advmil_mha_bwd, csrc/attn.hip) on the step's
                 region slab against the same bf16x3 MFMA roof; the dominant contraction is kept beside it as `gemm_roofline`.
  pool_roofline: the attention-pool call against the HBM roof, at the step slab and at one bag.
  sizes        : (1 GPU) the same step at the other sizes north_star names -- ABMIL 1k / 32k patches, ESAT 8k / 32k patches
                 (bags/s, and for ESAT the attention-core roofline) -- plus the exact-fp32 arithmetic mode and one bag per step.
  product_loop : (1 GPU) the product loop itself -- MyHandler._train_each_epoch, eager launches, ragged pinned host bags -- over
                 two epochs of the same loader: epoch 1 through the staging slab (PCIe-inclusive), epoch 2 out of the
                 device-resident bag cache (no PCIe); plus `graph_resident_split_in_step` (graph replay with fp32-only residency:
                 the operand-plane split inside every step), so the gap between `value` and the product loop can be apportioned
                 between PCIe, the per-step split and eager launch overhead; and `eval_loop`: the per-epoch evaluation
                 (MyHandler.test_model, reference model_handler.py:278-285 / 598-643) per bag on pageable host bags (the reference's
                 loop shape), batched over PCIe, and batched out of the bag cache.
  cpu_baseline : the oracle (pure PyTorch CPU restatement of the reference schedule, pinned against the
                 reference to <=1e-6) timed on this box's host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0 # ibid.: dense bf16 MFMA. The bf16x3 arithmetic issues 3 bf16 MFMAs per fp32-equivalent product,
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0   # so its roof in ALGORITHMIC flops (2*M*N*K) is a third of that


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=16, help="bags per optimizer step per GPU (bp_every_batch)")
    ap.add_argument("--patches", type=int, default=8192)
    ap.add_argument("--mode", default="abmil", choices=["abmil", "patch", "cluster", "graph"])
    ap.add_argument("--pool", type=int, default=64, help="distinct resident bags per GPU")
    ap.add_argument("--gemm-mode", default="bf16x3", choices=["bf16x3", "exact"],
                    help="arithmetic of the fp32 contraction engine: bf16x3 = split-bf16 products on the bf16 matrix pipe with fp32 "
                         "accumulate (near-fp32, parity-tested); exact = fp32 MFMA")
    ap.add_argument("--eager", action="store_true", help="drive the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-extras", "--no-bf16-extra", dest="no_extras", action="store_true",
                    help="skip the extra measurements (other sizes / backbones, exact-fp32 mode, one bag per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg (N > 1)")
    ap.add_argument("--cpu-bags", type=int, default=16, help="bags per optimizer step of the CPU-baseline sample (16 = the GPU line's step batch)")
    return ap.parse_args()


def make_pool(torch, dev, mode, n_pool, n_patches, seed, group=16, x_storage="fp32"):
    """Synthetic bags x ~ N(0,1) [1,N,1024] fp32 generated on the device + labels (t~U, e = i mod 2). The pool is laid out
    as slabs of `group` bags (what a loader's staging buffer looks like), so a step batch is one contiguous [group*N, 1024]."""
    g = torch.Generator(device=dev).manual_seed(seed)
    xs, ys, ys_host = [], [], []
    slab = None
    for i in range(n_pool):
        if i % group == 0:
            slab = torch.empty(min(group, n_pool - i), n_patches, 1024, device=dev)
            slab.normal_(generator=g)
            if x_storage == "bf16":               # the same numbers, held as ONE bf16 plane per bag (cfg x_storage = 'bf16')
                slab = slab.to(torch.bfloat16)
        x = slab[i % group].unsqueeze(0)
        if mode == "cluster":
            ext = torch.randint(0, 8, (1, n_patches), device=dev, generator=g).float()
        elif mode == "graph":   # patches on a sqrt(N) grid, 8-NN (tools/patchgcn_graph_s2.py:66-80 layout)
            from types import SimpleNamespace
            from advmil_amd import synth
            if i == 0:
                make_pool.ei = torch.from_numpy(synth.grid_knn_graph(n_patches, 8)).to(dev)
            ext = SimpleNamespace(x=x[0], edge_index=make_pool.ei)
        else:
            ext = torch.zeros(1, 1, device=dev)
        t = 0.05 + 0.9 * float(torch.rand((), device=dev, generator=g))
        y = torch.tensor([[t, float(i % 2)]], dtype=torch.float32)
        xs.append([x, ext]); ys_host.append(y); ys.append(y.to(dev))
    return xs, ys, ys_host


def cpu_baseline(args, torch):
    """Oracle train_step on a bounded sample (cpu-bags bags of the same size, shipped dropout rates as explicit masks
    drawn inside the timed region, the way the reference draws them)."""
    from oracle import advmil_oracle as O
    from advmil_amd.config import default_cfg
    from advmil_amd.model import Generator, load_backbone  # noqa: F401  (shapes only, built on CPU)
    from types import SimpleNamespace
    import advmil_amd.model.GANSurv as GS
    from advmil_amd.ingest import effective_cpus
    nthreads = min(torch.get_num_threads(), effective_cpus())      # (a cgroup CPU quota below the thread count only adds contention)
    torch.set_num_threads(nthreads)
    kind, N, nb = args.mode, args.patches, args.cpu_bags
    dg = 128 if kind == "graph" else 384
    bb = load_backbone(kind, [1024, dg, dg])
    G = Generator(dg, 1, bb, SimpleNamespace(noise=[0, 1], hops=1, noise_dist="uniform"), False, 0.6, "sigmoid")
    from advmil_amd.model.model_utils import init_weights
    G.apply(init_weights)
    ax = SimpleNamespace(in_dim=1024, out_dim=128, ksize=1, backbone="avgpool", dropout=0.25)
    ay = SimpleNamespace(in_dim=1, hid_dims=[64, 128], norm=False, dropout=0.0)
    D = GS.PrjDiscriminator(ax, ay, prj_path="x", inner_product="instance")
    PG = {k: v.detach().clone() for k, v in G.state_dict().items()}
    PD = {k: v.detach().clone() for k, v in D.state_dict().items()}
    gen = torch.Generator().manual_seed(0)
    from advmil_amd import synth as _synth
    ei = torch.from_numpy(_synth.grid_knn_graph(N, 8)) if kind == "graph" else None
    bags = [(torch.randn(1, N, 1024, generator=gen),
             (torch.randint(0, 8, (N,), generator=gen).float() if kind == "cluster" else ei),
             torch.tensor([[0.3 + 0.05 * i, float(i % 2)]])) for i in range(nb)]
    L = N // 16

    def drop(shape, p):
        return (torch.rand(shape) >= p).float() / (1 - p)

    def masks_g():
        if kind == "abmil":
            m = {"fc": drop((N, 384), .25), "att_a": drop((N, 384), .25), "att_b": drop((N, 384), .25), "rho": drop((1, 384), .25)}
        elif kind == "cluster":
            m = {"fc": drop((8, 384), .25), "att_a": drop((8, 384), .25), "att_b": drop((8, 384), .25)}
        elif kind == "graph":
            m = {"fc": drop((N, 128), .25), "phi": drop((N, 128), .25), "att_a": drop((N, 128), .25), "att_b": drop((N, 128), .25)}
        else:
            m = {"attn": drop((1, 8, L, L), .25), "drop1": drop((1, L, 384), .25), "ffn": drop((1, L, 384), .25),
                 "drop2": drop((1, L, 384), .25), "pool_a": drop((1, L, 384), .25), "pool_b": drop((1, L, 384), .25)}
        m["mlp0"] = drop((1, dg // 2), .6)
        return m

    def masks_d():
        return {"fc1": drop((1, L, 64), .25), "pool_a": drop((1, L, 128), .25), "pool_b": drop((1, L, 128), .25), "fc2": drop((1, 64), .25)}

    cfg = O.StepConfig(kind=kind)

    def one_step():
        nd = [[torch.rand(1, dg // 2)] for _ in range(nb)]
        ng = [[torch.rand(1, dg // 2)] for _ in range(nb)]
        O.train_step(cfg, PG, PD, {}, {}, bags, nd, ng, [masks_d() for _ in range(nb)], [masks_d() for _ in range(nb)],
                     [masks_g() for _ in range(nb)])

    full, bags = bags, bags[:min(4, nb)]
    nb_full, nb = nb, len(bags)
    one_step()                                   # warm-up on 4 bags (allocator, thread pool)
    bags, nb = full, nb_full
    t0 = time.perf_counter()
    reps = 0
    while True:
        one_step()
        reps += 1
        if time.perf_counter() - t0 > 12.0 or reps >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(nb * reps / dt, 4), "unit": "bags/s", "cores": nthreads, "kind": "port",
            "sample": f"{reps} optimizer step(s) of {nb} bags x {N} patches x 1024 fp32, {kind}+RLIP, shipped dropout rates, "
                      f"oracle/advmil_oracle.py::train_step, torch {torch.__version__} CPU, {nthreads} threads of {os.cpu_count()} cpus "
                      f"({nb} bags per optimizer step, as on the GPU line)"}


class Case:
    """One workload on this rank: a handler, its resident bag pool and one captured HIP graph per group of `bags` bags."""

    def __init__(self, torch, dev, mode, patches, bags, pool, gemm_mode, seed, eager=False, world=1, resident_planes=True, x_storage="fp32"):
        from advmil_amd.config import default_cfg
        from advmil_amd.model import MyHandler
        self.torch, self.dev, self.mode, self.patches, self.bags, self.world = torch, dev, mode, patches, bags, world
        cfg = default_cfg(bcb_mode=mode, bp_every_batch=bags, cuda_id=dev.index, gemm_mode=gemm_mode, x_storage=x_storage)
        if mode == "graph":            # PatchGCN dims of the reference's model_stats.py:63
            cfg.update(bcb_dims="1024-128-128", gen_dims="128-1")
        self.h = MyHandler(cfg, device=dev)
        self.h.resident_planes = resident_planes
        self.n_pool = max(pool, bags)
        self.xs, self.ys, self.ys_host = make_pool(torch, dev, mode, self.n_pool, patches, seed=seed, group=bags, x_storage=x_storage)
        self.cursor = 0
        self.graphs = []
        self.launch_note = "eager"
        if not eager:
            from advmil_amd.graphed import GraphedStep
            try:
                for g0 in range(0, self.n_pool - bags + 1, bags):
                    idx = list(range(g0, g0 + bags))
                    self.graphs.append(GraphedStep(self.h, [self.xs[i] for i in idx], [self.ys[i] for i in idx],
                                                   [self.ys_host[i] for i in idx], warmup=1))
                self.launch_note = (f"hipGraph replay ({len(self.graphs)} captured bag groups"
                                    + ((", ONE graph per step: the 2 all-reduces captured inside (ADVMIL_GRAPH_COLLECTIVES=1))"
                                        if self.graphs[0].captured_collectives else
                                        ", 4 segments: the 2 all-reduces stay outside capture, D's overlaps the G-forward segment)") if world > 1 else ")"))
            except Exception as exc:          # never lose the run to a capture problem: the eager schedule is the same step
                self.graphs = []
                torch.cuda.synchronize()
                self.launch_note = f"eager (graph capture failed: {type(exc).__name__}: {str(exc)[:120]})"

    def eager_step(self):
        i0, nb, h = self.cursor, self.bags, self.h
        idx = [(i0 + j) % self.n_pool for j in range(nb)]
        self.cursor = (i0 + nb) % self.n_pool
        bx, by, bh = [self.xs[i] for i in idx], [self.ys[i] for i in idx], [self.ys_host[i] for i in idx]
        plan = h._plan(bx, by, "wlabel", None, bh)
        h._update_disc(0, bx, by, "wlabel", None, ys_host=bh, plan=plan)
        h._update_gen(0, bx, by, "wlabel", None, ys_host=bh, plan=plan)
        h.rng.advance(1)
        if len(h.history) > 64:
            h.history.clear()

    def step(self):
        if not self.graphs:
            return self.eager_step()
        self.graphs[self.cursor % len(self.graphs)].replay()
        self.cursor += 1

    def timed(self, steps, warmup, barrier):
        for _ in range(warmup):
            self.step()
        beat("warm-up enqueued")
        barrier()
        beat("timed region")
        t0 = time.perf_counter()
        for k in range(steps):
            self.step()
            if (k & 15) == 15:
                beat(f"timed step {k + 1} of {steps} enqueued")
        t_submit = time.perf_counter() - t0     # host time to enqueue all K steps (== dt when the host is the bottleneck)
        barrier()
        return time.perf_counter() - t0, t_submit

    def logs_finite(self):
        h = self.h
        logs = h.pop_logs() if not self.graphs else [h.resolve_log(d) for g in self.graphs for d in g.logs]
        return all(v == v and abs(v) != float("inf") for d in logs for k, v in d.items() if k != "i_batch")

    def free(self):
        self.graphs, self.xs, self.ys, self.h = [], None, None, None
        import gc
        gc.collect()
        self.torch.cuda.empty_cache()


def in_step_launch_us(torch, ops, case, replays=8):
    """Durations of the step's large launches WHERE THEY RUN IN THE STEP: one more bag group is captured with device wall-clock stamps
    (ops.Stamps: one-thread kernels, graph nodes) before and after every slab-sized contraction / attention call, the graph is replayed
    `replays` times and the stamps are read after each replay. -> {(name, shape): {"us": mean, "n": launches per step, "flops": per launch}}.
    A stamp pair adds the two inter-kernel gaps (a few us) to the launch it brackets: the figure is an upper bound of the kernel's time."""
    from advmil_amd.graphed import GraphedStep
    h = case.h
    st = ops.Stamps(case.dev)
    idx = list(range(case.bags))
    ops.STAMPS = st
    try:
        g = GraphedStep(h, [case.xs[i] for i in idx], [case.ys[i] for i in idx], [case.ys_host[i] for i in idx], warmup=1)
    finally:
        ops.STAMPS = None
    per = len(st.tags) // 2                              # marks of one step (the warm-up step made the first half)
    acc = {}
    for k in range(replays + 2):
        g.replay()
        torch.cuda.synchronize()
        if k >= 2:
            for name, shape, flops, us in st.durations_us(start=len(st.tags) - per):
                a = acc.setdefault((name, shape), {"flops": flops, "us": []})
                a["us"].append(us)
    h.history.clear()
    del g
    return {k: {"flops": v["flops"], "us": sum(v["us"]) / len(v["us"]), "n": len(v["us"]) // replays, "series": v["us"]} for k, v in acc.items()}


def event_time_us(torch, fn, iters, warm=3):
    """Average duration of fn() over `iters` back-to-back calls between two HIP events on the launch stream (queue kept full)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def attention_roofline(torch, ops, dev, L, bags, p=0.25, iters=20, in_step=None):
    """The fused ESAT attention core on the step's region slab (bags x L tokens, 8 heads x 48): forward and backward launches
    timed back to back. Algorithmic flops: forward 4 L^2 d per bag (QK^T, PV), backward 10 L^2 d (five contractions), d = 384.
    Roof: dense bf16 MFMA / 3 (three bf16 MFMAs per fp32-equivalent product)."""
    d, nh = 384, 8
    qkv = torch.randn(bags * L, 3 * d, device=dev, requires_grad=True)
    go = torch.randn(bags * L, d, device=dev)
    seg = ops.Segments([L] * bags, dev)
    rng = ops.DeviceRng(dev, seed=1)
    with torch.no_grad():
        us_f = event_time_us(torch, lambda: ops.mha(qkv, nh, p, rng, seg=seg), iters)
    o = ops.mha(qkv, nh, p, rng, seg=seg)

    def bwd():
        qkv.grad = None
        o.backward(go, retain_graph=True)

    us_b = event_time_us(torch, bwd, iters)
    ff, fb = 4.0 * L * L * d * bags, 10.0 * L * L * d * bags
    b2b = {"fwd_launch_us": round(us_f, 1), "bwd_launches_us": round(us_b, 1), "achieved": round((ff + fb) / (us_f + us_b) / 1e6, 2),
           "frac": round((ff + fb) / (us_f + us_b) / 1e6 / PEAK_BF16X3_TFLOPS, 4), "launches": iters,
           "note": "the same launches repeated with nothing in between: the chip clocks down under the sustained matrix-pipe power draw, a regime the step never reaches"}
    method = f"{iters} back-to-back launches between two HIP events on the launch stream"
    if in_step:      # the step's own attention calls, stamped inside a captured step (in_step_launch_us): forward = mean of the eval (no dropout) and the training (dropout) pass
        f_ = [v for k, v in in_step.items() if k[0] == "mha_fwd" and k[1][0] == bags * L]
        b_ = [v for k, v in in_step.items() if k[0] == "mha_bwd" and k[1][0] == bags * L]
        if f_ and b_:
            us_f, us_b = f_[0]["us"], b_[0]["us"]
            method = ("device wall-clock stamps (one-thread kernels = graph nodes) around the step's own ops.mha forward (eval + training pass, mean) and "
                      "backward INSIDE a captured step, 8 replays: timed between their real neighbours, at the clock the step runs at")
    ach = (ff + fb) / (us_f + us_b) / 1e6
    one = ops.mha_bwd_single_pass(L, 48)
    return {"bound": "mfma", "kernel": "split_planes + attn_fwd_kernel<48,drop> (csrc/attn.hip) + attn_bwd_prep + " +
                                       ("attn_bwd_one_kernel + attn_dq_reduce_kernel (csrc/attn_bwd1.hip: single-pass backward)" if one
                                        else "attn_bwd_dq_kernel + attn_bwd_dkv_kernel (csrc/attn.hip: two-launch backward)"),
            "backward_form": "one" if one else "two",
            "back_to_back": b2b,
            "tokens_per_bag": L, "bags": bags, "heads": nh, "head_dim": 48, "attn_dropout": p,
            "achieved": round(ach, 2), "peak": round(PEAK_BF16X3_TFLOPS, 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16X3_TFLOPS, 4),
            "peak_note": "bf16x3: 3 bf16 MFMAs per fp32-equivalent product -> roof = dense bf16 MFMA peak / 3 in algorithmic flops",
            "traffic": None, "fwd_launch_us": round(us_f, 1), "fwd_tflops": round(ff / us_f / 1e6, 1),
            "bwd_launches_us": round(us_b, 1), "bwd_tflops": round(fb / us_b / 1e6, 1),
            "flops_per_launch": {"fwd": ff, "bwd": fb},
            "algorithmic_bytes_per_launch": {"fwd": 4.0 * bags * L * (4 * d + nh), "bwd": 4.0 * bags * L * (9 * d + 2 * nh)},
            "method": method}


def pool_roofline(torch, ops, dev, patches, bags, iters=40, in_step=None):
    Dh = 384
    nrows = bags * patches
    nbuf = max(2, int(600e6 // (4 * nrows * Dh)) + 1)          # rotate slabs past the 256 MB Infinity Cache
    hs = [torch.randn(nrows, Dh, device=dev) for _ in range(nbuf)]
    sc = torch.randn(nrows, device=dev)
    seg_b = ops.Segments([patches] * bags, dev)
    k = [0]

    def call():
        ops.softmax_pool(sc, hs[k[0] % nbuf], nrows, Dh, seg_b)
        k[0] += 1

    usp = event_time_us(torch, call, iters)
    byt = 4.0 * nrows * Dh + 3 * 4.0 * nrows
    ist = None
    if in_step:        # the step's own pooling calls (generator eval pass + training pass), stamped inside a captured step (in_step_launch_us)
        hits = [v for k, v in in_step.items() if k[0] == "softmax_pool_fwd" and k[1][0] == nrows and k[1][1] == Dh]
        if hits:
            us_i = sum(v["us"] * v["n"] for v in hits) / sum(v["n"] for v in hits)
            per = None
            if len(hits) == 1 and hits[0]["n"] == 2:      # program order inside a step: the generator's eval pass (D update), then its training pass
                ser = hits[0]["series"]
                ev_, tr_ = ser[0::2], ser[1::2]
                per = {"eval_pass_us": round(sum(ev_) / len(ev_), 2), "train_pass_us": round(sum(tr_) / len(tr_), 2)}
                per["eval_pass_frac"] = round(byt / per["eval_pass_us"] / 1e3 / 8000.0, 4)
                per["train_pass_frac"] = round(byt / per["train_pass_us"] / 1e3 / 8000.0, 4)
                per["note"] = ("the training pass's call runs right behind the gate contraction that STORES 403 MB of activations: it pays the write-back of "
                               "those dirty Infinity-Cache lines (tools/probe/pool_instep.py; DESIGN.md 4.3c); the eval pass's call has no such neighbour")
            ist = {"avg_call_us": round(us_i, 2), "calls_per_step": sum(v["n"] for v in hits), "achieved": round(byt / us_i / 1e3, 1),
                   "frac": round(byt / us_i / 1e3 / 8000.0, 4), "per_call": per,
                   "method": "device wall-clock stamps around the step's own advmil_softmax_pool_fwd calls inside a captured step (both launches + "
                             "the two inter-kernel gaps): the call between its real neighbours, at the clock the step runs at"}
    return {"bound": "hbm", "in_step": ist, "kernel": "pool_partial8_online + pool_merge_online (advmil_softmax_pool_fwd, 2 launches: online-softmax partials, merge + A)",
            "rows": nrows, "bags": bags, "achieved": round(byt / usp / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(byt / usp / 1e3 / 8000.0, 4), "avg_call_us": round(usp, 2), "algorithmic_bytes_per_call": byt,
            "rotating_slabs": nbuf,
            "method": f"{iters} back-to-back calls between two HIP events, rotating slabs > 256 MB; per-kernel durations: profiles/ (rocprofv3 --kernel-trace)"}


def visible_gpu_count():
    """GPUs this process could use, counted WITHOUT touching HIP (on ROCm builds without amdsmi torch.cuda.device_count() calls
    hipGetDeviceCount, which initialises the runtime in the launcher process): the *_VISIBLE_DEVICES lists when set, else the KFD
    topology (nodes with SIMDs). The rank processes validate the count again when they set their device."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            try:
                props = dict(ln.split() for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) == 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        pass
    return n


class Watchdog:
    """Multi-rank runs only: a rank that makes no progress for `limit` seconds (a collective or a captured-graph mismatch between ranks
    hangs, it does not raise) prints where it stood and leaves with exit code 3 -- a fresh exit, never a re-exec; the launcher then
    tears the other ranks down. `beat(what)` is called at every phase boundary and every few steps of a timed loop."""

    def __init__(self, rank, limit=60.0):
        import threading
        self.rank, self.limit, self.last, self.what, self.on = rank, float(limit), time.time(), "start", True
        # until the graphs are captured a rank is in SETUP: first kernel loads, the resident pool's fill, RCCL's communicator coming up at the
        # first collective of the warm-up steps (tens of seconds on a fresh 8-GPU node) -- a longer leash there, or a healthy run gets shot
        self.setup_limit = max(self.limit, float(os.environ.get("ADVMIL_BENCH_SETUP_S", "300")))
        self.setup = True
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def beat(self, what, setup=None):
        """setup = True / False switches the leash: every Case construction (pool fill, warm-up steps with collectives, graph capture)
        and every rank-0-only section that the other ranks sit out in a barrier is SETUP; the timed legs are not."""
        self.last, self.what = time.time(), what
        if setup is not None:
            self.setup = bool(setup)

    def stop(self):
        self.on = False

    def _run(self):
        while self.on:
            time.sleep(1.0)
            lim = self.setup_limit if self.setup else self.limit
            if self.on and time.time() - self.last > lim:
                print(f"bench.py watchdog: rank {self.rank} made no progress for {lim:.0f} s (last: {self.what}); exiting 3",
                      file=sys.stderr, flush=True)
                os._exit(3)


WATCHDOG = None


def beat(what, setup=None):
    if WATCHDOG is not None:
        WATCHDOG.beat(what, setup)


def self_launch(args):
    """`bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a CHILD torch.distributed.run job (this
    process never touches the GPU: the devices are counted from the environment / the KFD topology, visible_gpu_count), relay rank
    0's line, exit with its code."""
    import socket
    import subprocess
    ndev = visible_gpu_count()
    backend = os.environ.get("ADVMIL_DIST_BACKEND")
    if 0 < ndev < args.gpus and backend != "gloo":            # (0 = could not tell: the ranks check when they set their device)
        print(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible: refusing to report a {args.gpus}-GPU number from fewer "
              "devices (set ADVMIL_DIST_BACKEND=gloo to run ranks that share devices, for functional testing only)", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # (own process group + an overall limit: a rank job that outlives its watchdogs is killed as a group, by the PIDs this process started)
    limit = float(os.environ.get("ADVMIL_BENCH_TIMEOUT", "1500"))
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, _ = child.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(child.pid, signal.SIGKILL)
        stdout, _ = child.communicate()
        print(f"bench.py: the {args.gpus}-rank job exceeded {limit:.0f} s and was killed", file=sys.stderr)
        sys.stdout.write(stdout or "")
        sys.exit(3)
    proc = SimpleNamespace(stdout=stdout or "", returncode=child.returncode)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the rank job printed no result line", file=sys.stderr)
        sys.exit(1)
    sys.exit(proc.returncode)


def xbf16_parity(torch, dev, args, steps=2):
    """Largest deviation of the x_storage = 'bf16' step from the fp32-storage step over `steps` eager optimizer steps on the same bags,
    weights, noise and dropout draws: logged losses, the collected predictions / discriminator scores, and the post-step weights."""
    res = {}
    runs = {}
    for xs in ("fp32", "bf16"):
        c = Case(torch, dev, args.mode, args.patches, args.bags, args.bags * steps, args.gemm_mode, 777, eager=True, x_storage=xs)
        c.h.rng.reset(4242)
        outs = []
        for _ in range(steps):
            i0, nb, h = c.cursor, c.bags, c.h
            idx = [(i0 + j) % c.n_pool for j in range(nb)]
            c.cursor = (i0 + nb) % c.n_pool
            bx, by, bh = [c.xs[i] for i in idx], [c.ys[i] for i in idx], [c.ys_host[i] for i in idx]
            plan = h._plan(bx, by, "wlabel", None, bh)
            preds, fakes = h._update_disc(0, bx, by, "wlabel", None, ys_host=bh, plan=plan)
            h._update_gen(0, bx, by, "wlabel", None, ys_host=bh, plan=plan)
            h.rng.advance(1)
            outs.append((torch.cat(preds).detach().double().cpu(), torch.cat(fakes).detach().double().cpu()))
        logs = c.h.pop_logs()
        runs[xs] = (outs, logs, c.h.optimizerG.flat_param.detach().double().cpu(), c.h.optimizerD.flat_param.detach().double().cpu())
        c.free()
        del c
    (oa, la, ga, da), (ob, lb, gb, db) = runs["fp32"], runs["bf16"]
    res["y_hat_maxabs"] = max(float((a[0] - b[0]).abs().max()) for a, b in zip(oa, ob))
    res["f_fake_maxabs"] = max(float((a[1] - b[1]).abs().max()) for a, b in zip(oa, ob))
    res["loss_maxabs"] = max(abs(float(x[k]) - float(y[k])) for x, y in zip(la, lb) for k in x if k != "i_batch")
    res["weights_G_maxabs"] = float((ga - gb).abs().max())
    res["weights_D_maxabs"] = float((da - db).abs().max())
    res["parity_maxabs"] = max(res["y_hat_maxabs"], res["f_fake_maxabs"], res["loss_maxabs"])
    res["steps"] = steps
    res["note"] = ("north_star contract: 1e-4 on predictions, scores and losses. Post-Adam weights move by up to 2 lr where a gradient "
                   "component's sign flips (Adam's first steps are lr * sign(g)): lr_G = 8e-5")
    return res


def product_loop(args, torch, dev, case):
    from advmil_amd.config import default_cfg
    from advmil_amd.graphed import GraphedStep
    from advmil_amd.model import MyHandler
    if case is not None and case.h is not None:
        case.free()
    out = {}
    hh = MyHandler(default_cfg(bcb_mode=args.mode, bp_every_batch=args.bags, cuda_id=dev.index, gemm_mode=args.gemm_mode), device=dev)
    gcpu = torch.Generator().manual_seed(7)
    nsteps, base = (30 if args.patches <= 8192 else 8), args.patches
    fr = (0.75, 1.0, 1.25, 0.5, 1.5, 1.0, 0.875, 1.125)                      # ragged: 0.5x .. 1.5x the nominal size
    nbag = args.bags * (nsteps + 2)
    # (minus a few 16-row regions per bag: the rows of a real step batch are a multiple of 16, not of the slab kernels' 256-row tiles --
    # the loop pads the staged slab with zero rows, ingest.SlabStager.pad_rows)
    lens = [int(base * fr[i % len(fr)]) // 16 * 16 - 16 * ((i * 7) % 11) for i in range(nbag)]
    distinct = min(nbag, 64 if args.patches <= 8192 else 16)                   # distinct host bags (pinned): 64 x 33.5 MB = 2.1 GB
    hostpool = [torch.randn(1, lens[i], 1024, generator=gcpu).pin_memory() for i in range(distinct)]
    # the loader's patient index IS the cache key: bag i of the epoch is patient i mod `distinct`
    loader = [(torch.tensor([[i % distinct]], dtype=torch.int), [hostpool[i % distinct], torch.zeros(1, 1)],
               torch.tensor([[0.3 + 0.01 * (i % 50), float(i % 2)]])) for i in range(nbag)]
    rows = sum(hostpool[i % distinct].shape[1] for i in range(2 * args.bags, nbag))
    os.environ["ADVMIL_BAG_CACHE_GB"] = "0"
    hh._train_each_epoch(loader[:2 * args.bags], "warmup")                     # allocates the pinned + device staging slabs
    torch.cuda.synchronize()

    # a DataLoader-like object: the bag cache's scope is the loader's `.dataset` (a bare list is re-read every epoch unless
    # cfg['bag_cache_gb'] asks for the cache explicitly)
    class _DS:
        def __init__(self, items):
            self.items = items

    class _DL:
        def __init__(self, ds):
            self.dataset = ds

        def __iter__(self):
            return iter(self.dataset.items)

    train_dl = _DL(_DS(loader[2 * args.bags:]))

    def epoch(tag):
        t1 = time.perf_counter()
        hh._train_each_epoch(train_dl, tag)
        torch.cuda.synchronize()
        return time.perf_counter() - t1

    def ent(dte, path):
        return {"value": round(args.bags * nsteps / dte, 2), "unit": "bags/s", "ms_per_step": round(1e3 * dte / nsteps, 3), "steps": nsteps,
                "patches_per_bag": "ragged %d..%d (mean %d)" % (min(lens), max(lens), rows // (args.bags * nsteps)), "path": path}

    d0 = epoch("nocache")
    out["eager_pcie_ragged"] = ent(d0, "MyHandler._train_each_epoch, eager launches, pinned host bags -> SlabStager (copy stream) -> step slab; "
                                       "bag cache off: every epoch pays PCIe")
    out["eager_pcie_ragged"]["h2d_mb_per_step"] = round(rows * 4096 / nsteps / 1e6, 1)
    # the same loop fed what a default DataLoader (pin_memory=False, the reference's) hands over: pageable tensors, copied into the
    # pinned staging slab by the ingest's own thread pool (advmil_amd/ingest.py::host_copy_rows) on their way to the DMA
    npg = min(nsteps, 8)
    pageable = [(it[0], [it[1][0].clone(), it[1][1]], it[2]) for it in loader[2 * args.bags:(2 + npg) * args.bags]]
    hh._train_each_epoch(pageable[:args.bags], "warmup")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    hh._train_each_epoch(pageable, "nocache")
    torch.cuda.synchronize()
    dpg = time.perf_counter() - t1
    out["eager_pcie_pageable_ragged"] = {"value": round(args.bags * npg / dpg, 2), "unit": "bags/s", "ms_per_step": round(1e3 * dpg / npg, 3), "steps": npg,
                                         "path": "same loop, bag cache off, PAGEABLE host bags (DataLoader without pin_memory): pageable -> pinned slab by a "
                                                 "pool of 8 copy threads -> DMA"}
    del pageable
    del os.environ["ADVMIL_BAG_CACHE_GB"]
    hh._bag_caches = {}
    d1 = epoch("train")                                                          # epoch 1: PCIe + fills the cache
    cache = hh._bag_caches.get("train")
    st1 = None if cache is None else cache.stats()
    d2 = epoch("train")                                                          # epoch 2: out of HBM
    out["epoch1_fill_cache"] = ent(d1, "same loop, device-resident bag cache on (default), cache empty at the start: a bag's FIRST visit pays PCIe + one "
                                       "D2D copy + its plane split, later visits are hits (this synthetic epoch walks its %d distinct patients "
                                       "%.1f times, see `cache`)" % (distinct, (nbag - 2 * args.bags) / distinct))
    out["epoch1_fill_cache"]["cache"] = st1
    out["epoch2_resident_capturing"] = ent(d2, "same loop, second epoch: every bag served from the HBM cache (no PCIe); the epoch's step-batch keys "
                                               "are new (cached bags are staged as operand planes only): first sight eager, second sight = capture of "
                                               "the key's step graph (its cost is in this epoch), later batches replayed")
    sg0 = dict(hh.step_graph_stats)
    d3 = epoch("train")                                                          # epoch 3: steady state of a long run
    sg1 = dict(hh.step_graph_stats)
    out["resident_ragged"] = ent(d3, "same loop, third epoch (steady state): every bag from the HBM cache, step slab + operand planes assembled by D2D "
                                     "copies on the copy stream under the previous step; every step batch whose key (bags, padded slab rows, real "
                                     "pairs, slab buffer) has a captured graph is REPLAYED with its plan arrays rewritten (MyHandler step graphs), "
                                     "others run eagerly")
    out["resident_ragged"]["cache"] = None if cache is None else cache.stats()
    out["resident_ragged"]["steps_replayed_captured_eager"] = [sg1[k] - sg0[k] for k in ("replayed", "captured", "eager")]
    out["resident_ragged"]["step_graphs_held"] = len(hh._step_graph_cache)
    # the same steady-state epoch with every step issued eagerly (what rounds 3-5 reported as the resident epoch)
    hh.step_graphs_max, keep = 0, (hh._step_graph_cache, hh.step_graphs_max)
    hh._step_graph_cache = {}
    d4 = epoch("train")
    hh._step_graph_cache, hh.step_graphs_max = keep
    out["eager_resident_ragged"] = ent(d4, "same epoch with every step issued eagerly (no step graph taken): ~53 launches per step from Python")

    # ---- the same loop with the bags held as ONE bf16 plane (cfg x_storage = 'bf16'): a loader that stores bf16 features hands over
    # half the bytes (PCIe, staging slab, device cache); fp32 host bags would cross PCIe as they are and be rounded on the copy stream
    if args.gemm_mode == "bf16x3":
        try:
            hb = MyHandler(default_cfg(bcb_mode=args.mode, bp_every_batch=args.bags, cuda_id=dev.index, gemm_mode=args.gemm_mode,
                                       x_storage="bf16"), device=dev)
            pool16 = [t.to(torch.bfloat16).pin_memory() for t in hostpool]
            items16 = [(it[0], [pool16[int(it[0].reshape(-1)[0])], it[1][1]], it[2]) for it in loader]
            dl16 = _DL(_DS(items16[2 * args.bags:]))
            os.environ["ADVMIL_BAG_CACHE_GB"] = "0"
            hb._train_each_epoch(items16[:2 * args.bags], "warmup")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hb._train_each_epoch(dl16, "nocache")
            torch.cuda.synchronize()
            e0 = time.perf_counter() - t1
            del os.environ["ADVMIL_BAG_CACHE_GB"]
            hb._bag_caches = {}
            t1 = time.perf_counter()
            hb._train_each_epoch(dl16, "train")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hb._train_each_epoch(dl16, "train")
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            out["x_storage_bf16"] = {
                "eager_pcie_ragged": dict(ent(e0, "bf16 host bags (pinned) -> bf16 staging slab: every epoch over PCIe, half the bytes"),
                                          h2d_mb_per_step=round(rows * 2048 / nsteps / 1e6, 1)),
                "eager_resident_ragged": dict(ent(e2, "second epoch out of the bf16 device cache (2 B per element resident)"),
                                              cache=None if hb._bag_caches.get("train") is None else hb._bag_caches["train"].stats())}
            del hb, pool16, items16, dl16
            from advmil_amd import ingest as _ing
            _ing.device_bag_cache(dev).clear()
        except Exception as exc:
            out["x_storage_bf16"] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
            os.environ.pop("ADVMIL_BAG_CACHE_GB", None)
            torch.cuda.synchronize()

    # ---- the per-epoch evaluation (reference `_run_training`, model_handler.py:278-285: MyHandler.test_model over the validation and
    # the test set after EVERY epoch, times_test_sample = 1; 598-643: one synchronous `.cuda()`, full forwards and 4 `.cpu()` syncs
    # per bag). Same ragged pinned host bags behind a DataLoader-like object (the cache scope is its `.dataset`).
    nev = min(len(loader), 8 * args.bags)
    ev_items = loader[:nev]

    def ev(ld, n, **kw):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = MyHandler.test_model(hh.netG, hh.netD, args.mode, ld, times_test_sample=1, **kw)
        torch.cuda.synchronize()
        assert r["y_hat"].shape[0] == n and bool(torch.isfinite(r["y_hat"]).all()), (r["y_hat"].shape, n)
        return time.perf_counter() - t1

    def eent(dte, path):
        return {"value": round(nev / dte, 2), "unit": "bags/s", "bags": nev, "ms_per_bag": round(1e3 * dte / nev, 3), "path": path}

    pageable = [(it[0], [it[1][0].clone(), it[1][1]], it[2]) for it in ev_items[:2 * args.bags]]        # what a default DataLoader hands over
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    MyHandler.test_model(hh.netG, hh.netD, args.mode, pageable, times_test_sample=1, batch_bags=1)
    torch.cuda.synchronize()
    d_ref = (time.perf_counter() - t1) / len(pageable) * nev
    ev(ev_items[:2 * args.bags], 2 * args.bags)                                        # (warm the evaluation's own staging slabs)
    dl = _DL(_DS(ev_items))
    d_b1 = ev(dl, nev)
    d_b2 = ev(dl, nev)
    out["eval_loop"] = {
        "per_bag_pageable": eent(d_ref, "MyHandler.test_model(batch_bags=1) on pageable host bags: the reference's loop shape -- synchronous copy, one "
                                        "forward of G and D per bag (backbone once per bag), results fetched per epoch; timed on %d bags" % len(pageable)),
        "batched_first_pass": eent(d_b1, "default: %d bags per slab through the training step's slab kernels, pinned staging slab on the copy stream; "
                                         "first pass over this dataset = PCIe + fills the device-resident bag cache" % int(os.environ.get("ADVMIL_EVAL_BATCH_BAGS", "16"))),
        "batched_resident": eent(d_b2, "same call, second pass (the next epoch's evaluation): every bag out of the HBM cache, no PCIe"),
    }
    del dl, pageable, ev_items
    del hh, hostpool, loader
    from advmil_amd import ingest as _ingest
    _ingest.device_bag_cache(dev).clear()
    import gc
    gc.collect(); torch.cuda.empty_cache()
    # graph replay with fp32-only residency: the bf16x3 operand planes of the slab are re-derived inside every step
    if args.gemm_mode == "bf16x3":
        c3 = Case(torch, dev, args.mode, args.patches, args.bags, min(args.pool, 32 if args.patches <= 8192 else 16), args.gemm_mode, 777,
                  resident_planes=False)
        n3 = 40 if args.patches <= 8192 else 10
        d3, _ = c3.timed(n3, 2, torch.cuda.synchronize)
        out["graph_resident_split_in_step"] = {"value": round(args.bags * n3 / d3, 2), "unit": "bags/s", "ms_per_step": round(1e3 * d3 / n3, 3),
                                               "steps": n3, "launch": c3.launch_note,
                                               "path": "HIP-graph replay over resident fp32 bags WITHOUT resident operand planes: "
                                                       "advmil_split_planes of the step slab runs inside every step (half the resident bytes)"}
        c3.free()
    return out


def genconv_roofline(torch, ops, dev, patches, bags, iters=20):
    """GENConv softmax aggregation (csrc/graph.hip) on the step's block-diagonal graph (bags x patches nodes, 8-NN grid, C = 128)
    against the HBM roof. Algorithmic bytes (8N x 128 x 4 gathered + N x 128 x 4 written), per launch:
    forward = 8 neighbour rows + own row read, out / lse / agg written; backward = per out-edge (dout, lse, agg) rows of the
    target + own (x, dout) read, dx written (dt comes out of the same walk); index arrays 4 B per edge + 4 B per node."""
    from advmil_amd import synth
    C, N = 128, bags * patches
    ei1 = torch.from_numpy(synth.grid_knn_graph(patches, 8)).to(dev).long()
    ei = torch.cat([ei1 + b * patches for b in range(bags)], dim=1)
    csr = ops.GraphCSR(ei, N)
    E = int(ei.shape[1])
    t = torch.ones(1, device=dev, requires_grad=True)
    nbuf = max(2, int(600e6 // (4 * N * C)) + 1)
    xs = [torch.randn(N, C, device=dev, requires_grad=True) for _ in range(nbuf)]
    k = [0]

    def fwd():                       # the training forward: out + the two rows kept for the backward (lse, agg)
        ops.genconv_aggregate(xs[k[0] % nbuf], t, csr)
        k[0] += 1

    def fwd_eval():                  # evaluation: `out` only
        with torch.no_grad():
            ops.genconv_aggregate(xs[k[0] % nbuf], t, csr)
        k[0] += 1

    us_f = event_time_us(torch, fwd, iters)
    us_fe = event_time_us(torch, fwd_eval, iters)
    ys = [ops.genconv_aggregate(x, t, csr) for x in xs]
    go = torch.randn(N, C, device=dev)
    L = ops._lib.lib()
    saved = [(y.grad_fn.saved_tensors, y.grad_fn) for y in ys]

    nws = L.advmil_genconv_bwd_workspace_bytes(N, C)
    ws, dt = torch.empty(nws // 4, device=dev), torch.empty(1, device=dev)
    dxs = [torch.empty(N, C, device=dev) for _ in range(2)]

    def bwd():                       # the edge walk + the one-workgroup sum of its dt partials (two launches, timed together)
        x, tt, lse, agg = saved[k[0] % nbuf][0]
        ops._lib.check(L.advmil_genconv_bwd(ops._p(go), ops._p(x), ops._p(agg), ops._p(lse), ops._p(csr.rowptr_src), ops._p(csr.col_dst),
                                            ops._p(tt), 1e-7, N, C, ops._p(dxs[k[0] % 2]), ops._p(dt), ops._p(ws), nws, ops._stream()),
                       "genconv_bwd")
        k[0] += 1

    us_b = event_time_us(torch, bwd, iters)
    row = 4.0 * C
    # compulsory HBM bytes: every row of every array once (a row gathered by several neighbouring targets is an L2 hit after its
    # first read); the per-EDGE figure of SURVEY 8d (8N x 128 x 4 gathered) is what the load path serves, reported as `gathered`
    bytes_f = N * row + 3 * N * row + 4.0 * (E + N)
    bytes_b = 4 * N * row + N * row + 4.0 * (E + N)
    gath_f, gath_b = E * row, 3.0 * E * row
    ach = (bytes_f + bytes_b) / (us_f + us_b) / 1e3
    return {"bound": "hbm", "kernel": "genconv_fwd128_kernel<save> + genconv_bwd128_kernel (+ genconv_dt_reduce_kernel) (csrc/graph.hip)", "nodes": N, "edges": E, "channels": C,
            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
            "fwd_launch_us": round(us_f, 1), "fwd_gbps": round(bytes_f / us_f / 1e3, 1), "bwd_launch_us": round(us_b, 1),
            "bwd_gbps": round(bytes_b / us_b / 1e3, 1), "algorithmic_bytes_per_launch": {"fwd": bytes_f, "bwd": bytes_b},
            "fwd_eval_launch_us": round(us_fe, 1), "fwd_eval_gbps": round((bytes_f - 2 * N * row) / us_fe / 1e3, 1),
            "gathered_bytes_per_launch": {"fwd": gath_f, "bwd": gath_b},
            "gathered_gbps": {"fwd": round((gath_f + bytes_f - N * row) / us_f / 1e3, 1), "bwd": round((gath_b + bytes_b - 4 * N * row) / us_b / 1e3, 1)},
            "note": "algorithmic = compulsory HBM bytes (each row of x / out / lse / dout once, outputs once, indices); `gathered` counts a "
                    "neighbour row once per EDGE (SURVEY 8d's 8N x 128 x 4): those re-reads hit in L2, so it is a load-path rate, not HBM",
            "rotating_inputs": nbuf, "method": f"{iters} back-to-back launches between two HIP events on the launch stream"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    import torch
    import torch.distributed as dist
    from advmil_amd import ops, parallel

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        if world > 1:
            args.gpus = world
        else:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}")
    ranks_seen = dist.get_world_size() if world > 1 else 1
    backend = dist.get_backend() if world > 1 else None
    global WATCHDOG
    if world > 1:
        WATCHDOG = Watchdog(rank, float(os.environ.get("ADVMIL_BENCH_STALL_S", "60")))
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        beat("barrier passed")

    def per_rank(x):
        """every rank's value of a scalar, in rank order (N > 1)"""
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [round(float(o.item()), 4) for o in outs]

    def exposed_allreduce(c, replays=6):
        """(D wait ms, G wait ms) per step of the segmented replay, mean over `replays` stamped replays, max over ranks: how long a rank's
        compute stream stood still for the two gradient exchanges (D's is started before the generator's forward and waited for behind it)."""
        if not c.graphs or len(c.graphs[0].segments) == 1:
            return None
        acc = [0.0, 0.0]
        for k in range(replays):
            g = c.graphs[k % len(c.graphs)]
            g.stamp_waits = True
            g.replay()
            torch.cuda.synchronize()
            d_ms, g_ms = g.exposed_allreduce_ms()
            g.stamp_waits = False
            acc[0] += d_ms / replays; acc[1] += g_ms / replays
        t = torch.tensor(acc, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"d_wait_ms": round(float(t[0]), 4), "g_wait_ms": round(float(t[1]), 4), "replays": replays,
                "note": "event-stamped waits of the 4-segment replay, max over ranks"}

    case = Case(torch, dev, args.mode, args.patches, args.bags, args.pool, args.gemm_mode, 1234 + rank, args.eager, world)
    beat("graphs captured", setup=False)
    if world > 1 and os.environ.get("ADVMIL_BENCH_TEST_STALL") == str(rank):      # self-test of the watchdog: this rank stops making progress
        time.sleep(1e6)
    ok = torch.tensor([1.0 if (case.graphs or args.eager) else 0.0], device=dev)
    if world > 1:                          # all ranks must take the same path
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        case.graphs = []
    h = case.h
    dt, t_submit = case.timed(args.steps, args.warmup, barrier)
    multi = None
    if world > 1:
        multi = {"per_rank_ms_per_step": [round(1e3 * v / args.steps, 4) for v in per_rank(dt)],
                 "collectives_in_graph": bool(case.graphs and case.graphs[0].captured_collectives)}
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        multi["exposed_allreduce"] = exposed_allreduce(case)
        beat("weak leg done")
    finite = case.logs_finite()

    # ---- strong scaling (N > 1): the reference's own optimizer step of 16 bags (cfg_nlst.yaml:71) split over the ranks, bag i of
    # the step batch on rank i mod W -> 16 / W bags per rank per step; same barrier / max-over-ranks timing
    strong = None
    GLOBAL_STEP = 16
    if world > 1 and not args.no_strong:
        if GLOBAL_STEP % world == 0:
            per = GLOBAL_STEP // world
            beat("strong leg: building the case", setup=True)      # a second pool fill + warm-up with collectives + capture: setup again
            cs = Case(torch, dev, args.mode, args.patches, per, max(per, min(args.pool, 8 * per)), args.gemm_mode, 99 + rank, args.eager, world)
            oks = torch.tensor([1.0 if (cs.graphs or args.eager) else 0.0], device=dev)
            dist.all_reduce(oks, op=dist.ReduceOp.MIN)
            if float(oks.item()) == 0.0:
                cs.graphs = []
            beat("strong leg: graphs captured", setup=False)
            dts, _ = cs.timed(args.steps, args.warmup, barrier)
            strong_ranks = [round(1e3 * v / args.steps, 4) for v in per_rank(dts)]
            ts = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            dts = float(ts.item())
            strong_wait = exposed_allreduce(cs)
            beat("strong leg done")
            strong = {"scaling": "strong", "value": round(GLOBAL_STEP * args.steps / dts, 3), "unit": "bags/s",
                      "per_rank_ms_per_step": strong_ranks, "exposed_allreduce": strong_wait,
                      "collectives_in_graph": bool(cs.graphs and cs.graphs[0].captured_collectives),
                      "ms_per_step": round(1e3 * dts / args.steps, 3), "steps": args.steps, "global_bags_per_step": GLOBAL_STEP,
                      "bags_per_step_per_gpu": per, "gd_steps_per_sec": round(args.steps / dts, 3),
                      "losses_finite": bool(cs.logs_finite()), "launch": cs.launch_note,
                      "note": "the reference's step batch (bp_every_batch = 16) split over the ranks; the same global optimizer step at every N"}
            cs.free()
            del cs
        else:
            strong = {"skipped": f"16 bags per step do not divide over {world} ranks"}

    # replicas must hold bit-identical weights after the timed steps (same reduced gradients, same Adam)
    in_sync = True
    if world > 1:
        # everything from here to the final barrier is untimed bookkeeping: the instrumented eager pass on every rank, then rank 0's own
        # roofline / cpu_baseline / extras legs (minutes) while the other ranks wait in the barrier -> the long leash again
        beat("timed legs done: roofline + extras", setup=True)
        cs = torch.stack([h.optimizerG.flat_param.double().sum(), h.optimizerD.flat_param.double().sum()])
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool(torch.equal(hi, lo))

    # ---- roofline of the dominant contraction.
    # (1) an instrumented eager pass over the same steps brackets every GEMM launch with HIP events on the launch
    #     stream to find which (kernel, shape) owns the most device time;
    # (2) that exact launch is then timed back-to-back (queue kept full, so no host gaps pollute the per-launch time)
    #     between two HIP events on the same stream. achieved = algorithmic FLOPs per launch / that duration.
    gemm_roof = None
    ist_all = None
    nprof = max(1, min(2, args.steps))
    # The instrumented pass drives full optimizer steps, which contain the two gradient all-reduces: EVERY rank has to run it
    # (rank 0 alone deadlocks the job at world > 1 -- observed: a 2-rank default-flag run hung until its 900 s timeout).
    if not args.no_roofline:
        if rank == 0:
            ops.KERNEL_PROFILE = []
        case.cursor = 0
        for _ in range(nprof):
            case.eager_step()
        torch.cuda.synchronize()
        h.history.clear()
        beat("instrumented eager pass done")
    if rank == 0 and not args.no_roofline:
        prof, ops.KERNEL_PROFILE = ops.KERNEL_PROFILE, None
        agg = {}
        for name, shape, flops, e0, e1 in prof:
            key = (name, shape)               # the kernel + tile ops.gemm actually launched
            a = agg.setdefault(key, {"ms": 0.0, "n": 0, "flops": flops})
            a["ms"] += e0.elapsed_time(e1); a["n"] += 1
        if os.environ.get("ADVMIL_BENCH_TABLE"):   # per-step GEMM table (launch, shape, count, ms) on stderr
            for (kn, sh), v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"  {kn:28s} M,N,K,splits={sh}  n/step={v['n'] / nprof:.1f}  ms/step={v['ms'] / nprof:.3f}  "
                      f"us/launch={1e3 * v['ms'] / v['n']:.1f}  TF={v['flops'] * v['n'] / v['ms'] / 1e9:.0f}", file=sys.stderr)
        (kname, shape), top = max(agg.items(), key=lambda kv: kv[1]["ms"])
        M, N, K, sp = shape
        planes_kernel = kname.startswith("gemm_nt_planes") or kname.startswith("gemm_tn_planes")
        if planes_kernel:                     # both operands arrive as bf16 planes (hi, lo): the same 4 bytes per element
            a_kc = b_kc = kname.startswith("gemm_nt_planes")      # (TN form: both [K, .])
        else:
            a_kc, b_kc = kname.split("<")[1].startswith("1"), kname.split("<")[1].split(",")[1].startswith("1")
        A = torch.randn((M, K) if a_kc else (K, M), device=dev)
        B = torch.randn((N, K) if b_kc else (K, N), device=dev)
        out = torch.empty(M, N, device=dev)
        kw = dict(a_planes=ops.split_planes(A), b_planes=ops.split_planes(B)) if planes_kernel else {}
        if not planes_kernel:
            kw["tile"] = int(kname.rstrip(">").split(",")[2]) * 10 + int(kname.rstrip(">").split(",")[3])
        elif sp > 1:
            kw["splits"] = sp
        replay_form = "ops.gemm: the same kernel, tile and operand form as the step's launch"
        if "two layers" in kname:             # the two-layer first-layer launch is replayed AS ITSELF (both outputs, relu | none split, h's planes)
            n1 = int(kname.rstrip(">").split(",")[2])
            W1, W2 = B[:n1].contiguous(), B[n1:].contiguous()
            p1, p2, apl = ops.split_planes(W1), ops.split_planes(W2), kw["a_planes"]
            b1, b2 = torch.randn(n1, device=dev), torch.randn(N - n1, device=dev)
            us = event_time_us(torch, lambda: ops.gemm_two_layers(A, apl, W1, p1, b1, 1, W2, p2, b2, 0, True), 16)
            replay_form = f"ops.gemm_two_layers: relu({n1} columns, planes emitted) | none({N - n1} columns), one launch"
        else:
            us = event_time_us(torch, lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, **kw), 16)
        flops = 2.0 * M * N * K
        us_b2b = us
        in_step = None
        if case.graphs and world == 1:        # the launch where it runs: stamped inside one more captured step (see in_step_launch_us);
            # single process only -- a captured step holds the gradient all-reduces, which rank 0 alone must never enter
            try:
                ist = ist_all = in_step_launch_us(torch, ops, case)
                hit = ist.get((kname, shape))
                if hit:
                    in_step = {"avg_launch_us": round(hit["us"], 2), "launches_per_step": hit["n"],
                               "others_us": {f"{k[0]} {list(k[1][:3])}": round(v["us"], 1)
                                             for k, v in sorted(ist.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"])[:10]}}
                    us = hit["us"]
            except Exception as exc:
                in_step = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
        achieved = flops / us / 1e6
        traffic, traffic_src = None, None
        try:      # PMC passes are separate runs (profiles/README.md); only a measurement of THIS kernel + shape + mode counts
            pmc_name = next(n for n in ("r06_pmc_gemm.json", "r05_pmc_gemm.json", "r04_pmc_gemm.json", "r03_pmc_gemm.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            ent = pmc.get("planes_kernel_launches" if planes_kernel else "launches", {}).get(f"{M}x{N}x{K}")
            same = ent and (ent.get("kernel", "").startswith(kname.split(",")[0].rstrip(">")) if planes_kernel else ent.get("kernel") == kname)
            if same and ent.get("gemm_mode") == args.gemm_mode and ent.get("hbm_bytes_per_launch"):
                traffic, traffic_src = ent["hbm_bytes_per_launch"], f"profiles/{pmc_name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel + shape)"
        except Exception:
            pass
        total_gemm_ms = sum(v["ms"] for v in agg.values()) / nprof
        if args.gemm_mode == "bf16x3":
            peak, peak_note = PEAK_BF16X3_TFLOPS, ("bf16x3: 3 bf16 MFMAs per fp32-equivalent product -> roof = dense bf16 MFMA peak / 3 "
                                                   "in algorithmic 2MNK flops (the fp32 MFMA roof would be 157.3)")
        else:
            peak, peak_note = PEAK_F32_MFMA_TFLOPS, "fp32 MFMA peak"
        gemm_roof = {"bound": "mfma", "kernel": kname, "shape_MNK": [M, N, K], "achieved": round(achieved, 2),
                     "peak": round(peak, 1), "peak_note": peak_note, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(us, 2), "flops_per_launch": flops,
                     "algorithmic_bytes_per_launch": 4.0 * (M * K + N * K + M * N),
                     "launches_per_step": top["n"] // nprof, "share_of_gemm_time_eager": round(top["ms"] / nprof / total_gemm_ms, 3),
                     "method": ("device wall-clock stamps (one-thread kernels = graph nodes) before and after the launch INSIDE a captured step, "
                                "8 replays: the launch timed between its real neighbours, at the clock the step runs at (includes the two "
                                "inter-kernel gaps)" if in_step and "avg_launch_us" in in_step else
                                "16 back-to-back launches between two HIP events on the launch stream"),
                     "in_step": in_step,
                     "back_to_back": {"avg_launch_us": round(us_b2b, 2), "frac": round(flops / us_b2b / 1e6 / peak, 4), "launches": 16,
                                      "note": "the same launch repeated with nothing in between: the chip clocks down under the sustained matrix-pipe "
                                              "power draw (DESIGN.md section 4), a regime the step never reaches"},
                     "replayed_as": replay_form,
                     "eager_event_bracketed_us": {f"{k[0]} {list(k[1][:3])}": round(1e3 * v["ms"] / v["n"], 1)
                                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
        del A, B, out

    roof = gemm_roof
    beat("gemm roofline done")
    if rank == 0 and not args.no_roofline and args.mode == "patch":
        try:
            roof = attention_roofline(torch, ops, dev, args.patches // 16, args.bags, in_step=ist_all)
        except Exception as exc:
            roof = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}

    # ---- the attention-pool kernels against the HBM roof (north star: ">= 50% of HBM3E roofline"): segmented softmax statistics +
    # weighted row sum + merge over the hidden rows h[sum N, 384]; algorithmic bytes = h read once + scores read twice + attention
    # weights written once; at the step slab and at one bag (bp_every_batch = 1).
    pool_roof = None
    beat("attention roofline done")
    if rank == 0 and not args.no_roofline and args.mode == "abmil":
        try:
            pool_roof = pool_roofline(torch, ops, dev, args.patches, args.bags, in_step=ist_all)
            pool_roof["one_bag"] = pool_roofline(torch, ops, dev, args.patches, 1)
        except Exception as exc:
            pool_roof = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}

    # ---- extras (single GPU): one bag per optimizer step in both arithmetic modes, the exact-fp32 step, the other sizes / backbone
    sizes = None
    exact_extra = bp1_extra = None
    if world == 1 and rank == 0 and not args.eager and case.graphs and not args.no_extras:
        from advmil_amd.graphed import GraphedStep

        def sync_barrier():
            torch.cuda.synchronize()

        def bp1(n1=200, per=1):
            npool = min(8 * per, case.n_pool // per * per)
            g1 = [GraphedStep(h, [case.xs[i + j] for j in range(per)], [case.ys[i + j] for j in range(per)],
                              [case.ys_host[i + j] for j in range(per)], warmup=1) for i in range(0, npool, per)]
            for k in range(8):
                g1[k % len(g1)].replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(n1):
                g1[k % len(g1)].replay()
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            return {"value": round(n1 / d1, 2), "unit": "G+D steps/s (%d bag%s of %d patches per step)" % (per, "" if per == 1 else "s", args.patches),
                    "ms_per_step": round(1e3 * d1 / n1, 3), "steps": n1, "bags_per_step": per}

        try:
            if args.bags > 1:
                bp1_extra = {args.gemm_mode: bp1()}
                # the per-rank step of the reference's 16-bag optimizer step split over 8 GPUs (cfg_nlst.yaml:71): 2 bags per
                # rank. Measured on ONE GPU, without the two gradient all-reduces -- the launch-latency-bound regime that the 8-GPU
                # strong-scaling leg runs in (DESIGN.md section 6 projects from it; no multi-GPU hardware was available to the build)
                if args.bags >= 2 and case.n_pool >= 2:
                    bp1_extra["strong_split_rank_step"] = bp1(per=2)
        except Exception as exc:
            bp1_extra = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
        if args.gemm_mode == "bf16x3":
            try:
                ops.set_gemm_mode("exact")
                n3 = max(10, args.steps // 4)
                g3 = []
                for g0 in range(0, case.n_pool - args.bags + 1, args.bags):
                    idx = list(range(g0, g0 + args.bags))
                    g3.append(GraphedStep(h, [case.xs[i] for i in idx], [case.ys[i] for i in idx], [case.ys_host[i] for i in idx], warmup=1))
                for k in range(args.warmup):
                    g3[k % len(g3)].replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(n3):
                    g3[k % len(g3)].replay()
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t1
                Mx = min(131072, args.patches * args.bags)
                A_, B_, o_ = torch.randn(Mx, 1024, device=dev), torch.randn(384, 1024, device=dev), torch.empty(Mx, 384, device=dev)
                usx = event_time_us(torch, lambda: ops.gemm(A_, B_, True, True, Mx, 384, 1024, out=o_), 20)
                tfx = 2.0 * Mx * 384 * 1024 / usx / 1e6
                exact_extra = {"value": round(args.bags * n3 / dt3, 3), "unit": "bags/s", "ms_per_step": round(1e3 * dt3 / n3, 3), "steps": n3,
                               "dtype": "f32 (v_mfma_f32_32x32x2_f32)",
                               "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel<1,1,2,2> exact", "shape_MNK": [Mx, 384, 1024],
                                            "achieved": round(tfx, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(tfx / PEAK_F32_MFMA_TFLOPS, 4), "avg_launch_us": round(usx, 2)}}
                del g3, A_, B_, o_
                if args.bags > 1 and isinstance(bp1_extra, dict) and "error" not in bp1_extra:
                    bp1_extra["exact"] = bp1()
            except Exception as exc:
                exact_extra = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
            finally:
                ops.set_gemm_mode(args.gemm_mode)
        # other sizes / the ESAT backbone (north_star: N in {1k, 8k, 32k}; configs[3] = ESAT at 32k patches)
        case.free()
        sizes = {}
        for tag, mode_, patches_, bags_, pool_, steps_ in (("abmil_1024", "abmil", 1024, 16, 128, 60), ("abmil_32768", "abmil", 32768, 16, 16, 12),
                                                          ("esat_8192", "patch", 8192, 16, 64, 30), ("esat_32768", "patch", 32768, 16, 16, 12),
                                                          ("patchgcn_4096", "graph", 4096, 16, 32, 30)):
            if mode_ == args.mode and patches_ == args.patches:
                continue
            try:
                c2 = Case(torch, dev, mode_, patches_, bags_, pool_, args.gemm_mode, 4321)
                d2, _ = c2.timed(steps_, 2, sync_barrier)
                ent = {"value": round(bags_ * steps_ / d2, 2), "unit": "bags/s", "ms_per_step": round(1e3 * d2 / steps_, 3), "steps": steps_,
                       "bags_per_step": bags_, "distinct_resident_bags": c2.n_pool, "losses_finite": c2.logs_finite(), "launch": c2.launch_note}
                ist2 = None
                if mode_ == "patch" and c2.graphs:
                    try:
                        ist2 = in_step_launch_us(torch, ops, c2)
                    except Exception:
                        ist2 = None
                c2.free()
                del c2
                if mode_ == "patch":
                    ent["roofline"] = attention_roofline(torch, ops, dev, patches_ // 16, bags_, iters=10, in_step=ist2)
                if mode_ == "graph":           # configs[4]'s backbone at a size one GPU steps through quickly; its sparse gather vs HBM
                    ent["roofline"] = genconv_roofline(torch, ops, dev, patches_, bags_)
                sizes[tag] = ent
            except Exception as exc:
                sizes[tag] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
                torch.cuda.synchronize()

        # ---- the X-only bf16 storage mode (configs[1] says "bf16"; SURVEY 8: "bf16-storage / fp32-accumulate perf mode"): bags held as
        # ONE bf16 plane (resident pool, cache, staging: half the bytes), weights / activations / gradients still hi + lo, so every
        # contraction over the slab forms its products with two MFMAs instead of three. Same workload as `value`; the deviation from the
        # fp32-storage step on the SAME bags, weights and draws is measured in this run (two eager optimizer steps each).
        if args.gemm_mode == "bf16x3" and args.mode in ("abmil", "patch"):
            try:
                cb = Case(torch, dev, args.mode, args.patches, args.bags, args.pool, args.gemm_mode, 1234 + rank, x_storage="bf16")
                db_, _ = cb.timed(args.steps, args.warmup, sync_barrier)
                ent = {"value": round(args.bags * args.steps / db_, 2), "unit": "bags/s", "ms_per_step": round(1e3 * db_ / args.steps, 3),
                       "steps": args.steps, "x_storage": "bf16 (one plane per bag: 2 B per element resident; two MFMAs per product on the slab "
                       "contractions)", "losses_finite": cb.logs_finite(), "launch": cb.launch_note}
                cb.free()
                del cb
                ent["parity_vs_fp32_storage"] = xbf16_parity(torch, dev, args)
                sizes[f"{args.mode}_{args.patches}_xbf16"] = ent
            except Exception as exc:
                sizes[f"{args.mode}_{args.patches}_xbf16"] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
                torch.cuda.synchronize()

    # ---- extra (single GPU): the PRODUCT loop -- MyHandler._train_each_epoch, eager launches, RAGGED bags arriving as pinned host
    # tensors. Two epochs over the same loader: epoch 1 goes through the staging slab (advmil_amd/ingest.py::SlabStager, PCIe-
    # inclusive) and fills the device-resident bag cache; epoch 2 is served from HBM (BagCache: no PCIe, the step's operand planes
    # are a row gather). One step plan per batch, no HIP graph (ragged segments change every step). Never `value`.
    epoch_extra = None
    if world == 1 and rank == 0 and not args.no_extras and args.mode in ("abmil", "patch"):
        try:
            epoch_extra = product_loop(args, torch, dev, case)
        except Exception as exc:
            epoch_extra = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
            torch.cuda.synchronize()

    cpu = None
    beat("extras done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only; at N > 1 the other ranks would idle in the barrier
        cpu = cpu_baseline(args, torch)

    if rank == 0:
        bags_total = args.bags * world * args.steps
        names = {"abmil": "ABMIL", "patch": "ESAT (DualTrans_HS)", "cluster": "DeepAttMISL", "graph": "PatchGCN"}
        out = {
            "metric": "WSI bags/sec (full G+D step)", "value": round(bags_total / dt, 3), "unit": "bags/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16x3 (fp32 operands split hi+lo, 3 bf16 MFMAs per product, fp32 accumulate; fp32 storage)"
                      if args.gemm_mode == "bf16x3" else "f32"), "data": "synthetic",
            "config": {"workload": f"{names[args.mode]}+AdvMIL(RLIP prj discriminator), {args.patches}-patch x 1024 fp32 bags "
                                   + ("(BASELINE.json configs[1] shape; " if (args.mode == "abmil" and args.patches == 8192) else
                                      ("(BASELINE.json configs[3] shape; " if (args.mode == "patch" and args.patches == 32768) else "("))
                                   + "fp32 storage, "
                                   + ("bf16x3 split arithmetic: >= the bf16 the config names, within 2e-5 of the fp32 reference)"
                                      if args.gemm_mode == "bf16x3" else "exact fp32 MFMA arithmetic)"),
                       "bags_per_step_per_gpu": args.bags, "global_bags_per_step": args.bags * world, "gen_updates": 1,
                       "distinct_resident_bags_per_gpu": case.n_pool, "parallelism": f"bag-parallel dp{world}", "dropout": "shipped rates",
                       "launch": case.launch_note},
            "gd_steps_per_sec": round(args.steps / dt, 3), "losses_finite": bool(finite), "replicas_in_sync": in_sync,
            "host_submit_ms_per_step": round(1e3 * t_submit / args.steps, 3),
            "roofline": roof, "gemm_roofline": (gemm_roof if roof is not gemm_roof else None), "pool_roofline": pool_roof,
            "cpu_baseline": cpu, "exact_f32_mfma_mode": exact_extra, "bp_every_batch_1": bp1_extra, "sizes": sizes,
            "product_loop": epoch_extra, "strong_scaling": strong, "ranks_seen": ranks_seen, "dist_backend": backend,
            "multi_gpu": multi,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        beat("result printed", setup=True)
        if rank != 0 and WATCHDOG is not None:
            WATCHDOG.stop()                # rank 0's untimed legs can outlast any leash: the launcher's ADVMIL_BENCH_TIMEOUT covers the wait
        dist.barrier()
        if WATCHDOG is not None:
            WATCHDOG.stop()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
