"""GPU: the handler's step on the edge cases of the path, against the oracle on the same seeded inputs (dropout off,
injected noise): ragged bags inside one step, steps without any event bag (no real pair), partially visible labels
('wolabel'), the hinge / wasserstein losses, the concat and bag-level discriminators, checkpoint save/resume, and
HIP-graph re-capture after a learning-rate change."""
import numpy as np
import pytest
import torch

from advmil_amd import synth
from advmil_amd.config import default_cfg
from oracle import advmil_oracle as O
from tests import helpers as H
from tests.test_parity_gpu import DEV, close, load_synth, zero_dropout

pytestmark = pytest.mark.gpu


def run_case(kind="abmil", lens=(256, 512, 128, 64), events=None, visible=None, mode="wlabel", steps=2, tol=2e-5,
             check_weights=True, bag_seed0=40, **cfg_over):
    from advmil_amd.model import MyHandler
    nb = len(lens)
    h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb, **cfg_over), device=DEV)
    prj = cfg_over.get("disc_prj_path", "x")
    dt = cfg_over.get("disc_type", "prj")
    PG = load_synth(h.netG, f"G-{kind}:")
    PD = load_synth(h.netD, "D-prj:" if dt == "prj" else "D-cat:")
    zero_dropout(h.netG); zero_dropout(h.netD)
    bags_all, loader = [], []
    for s in range(steps):
        for j, n in enumerate(lens):
            i = s * nb + j
            x = H.bag(bag_seed0 + i, max(512, max(lens)))[:, :n].contiguous()
            y = H.label(i)
            if events is not None:
                y[0, 1] = float(events[j])
            ext = H.T(synth.cluster_ids(0, bag_seed0 + i, n)) if kind == "cluster" else None      # DeepAttMISL: a cluster id per patch
            bags_all.append((x, ext, y))
            loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext if ext is not None else torch.zeros(1, 1)], y))
    nd = [[H.noise_tensor("var_d", i, 192)] for i in range(steps * nb)]
    ng = [[H.noise_tensor("var_g", i, 192)] for i in range(steps * nb)]
    h.noise_hook = lambda ph, i: [(nd if ph == "d" else ng)[i][0].to(DEV)]
    if visible is not None:
        h.patient_id["train"] = [str(i) for i in range(steps * nb)]
        h.patient_id["label_visible"] = [str(i) for i in range(steps * nb) if visible[i % nb]]
    cl = h._train_each_epoch(loader, "train", mode)
    logs = h.pop_logs()
    cfg = O.StepConfig(kind=kind, disc_type=dt, inner_product=cfg_over.get("disc_prj_iprd", "instance"), prj_path=prj,
                       loss_netD=cfg_over.get("loss_netD", "bce"))
    stG, stD, oPG, oPD = {}, {}, PG, PD
    for s in range(steps):
        sl = slice(s * nb, s * nb + nb)
        vis = None if visible is None else list(visible)
        oPG, oPD, lg, yh, ff, _, _ = O.train_step(cfg, oPG, oPD, stG, stD, bags_all[sl], nd[sl], ng[sl], visible=vis)
        d, g = logs[2 * s], logs[2 * s + 1]
        got = {k.split("/")[-1]: v for k, v in list(d.items()) + list(g.items())}
        for k in ("Loss_D", "D_real", "D_fake", "Loss_G_fake", "Loss_G_time", "Loss_G_total"):
            assert abs(got[k] - lg[k]) < tol, (s, k, got[k], lg[k])
        close(cl["y_hat"][sl], yh, tol); close(cl["f_fake"][sl], ff, tol)
    for net, P in (((h.netG, oPG), (h.netD, oPD)) if check_weights else ()):
        for k, v in net.state_dict().items():
            if k.endswith("pool.fc2.bias") or k.endswith("attention_c.bias") or (tol > 2e-5 and k in ("prj_layer.bias", "fc.bias")):
                continue      # parameters whose true gradient is exactly 0: Adam amplifies round-off to +-lr on both sides
            # two Adam steps, lr = 8e-5. Adam normalises the gradient, so an entry whose true gradient is at round-off level
            # moves by up to +-lr per step with a sign decided by round-off on BOTH sides: allow a handful of such entries
            # (< 0.01 %), none further than the two-step sign-flip bound; everything else must agree to 5e-5.
            diff = (v.cpu() - P[k]).abs()
            n_off = int((diff >= 5e-5).sum())
            assert n_off <= max(1, diff.numel() // 10000), (k, n_off, diff.numel(), float(diff.max()), (diff >= 5e-5).nonzero()[:12].tolist())
            assert float(diff.max()) < 2.05 * 8e-5 * steps, (k, float(diff.max()))
    return h


@pytest.mark.parametrize("kind", ["abmil", "patch", "cluster"])
def test_padded_slab_step_vs_oracle(kind):
    """Step batches of 5 552 rows (not a multiple of the slab kernels' 256-row tiles): the staging slab pads them with 80 zero rows,
    carried as a dummy bag -- two optimizer steps against the oracle, event / censored bags mixed, weights included."""
    h = run_case(kind=kind, lens=(2064, 1040, 1536, 912), events=(1, 0, 1, 1))
    assert h._stager.pad == 80 and h.slab_pad == 256


@pytest.mark.parametrize("kind", ["abmil", "patch"])
def test_ragged_bags_in_one_step(kind):
    run_case(kind=kind, lens=(256, 512, 128, 64, 208, 16))


@pytest.mark.parametrize("gemm_mode", ["exact", "bf16x3"])
def test_cluster_backbone_through_the_adversarial_step(gemm_mode):
    """DeepAttMISL through _update_disc / _update_gen against the oracle, two optimizer steps, ragged bags (the reference's own
    handler cannot run this combination on this torch: model/backbone.py:112 raises IndexError on the float cluster ids, so the
    oracle -- pinned on the backbone's forward by golden G1 -- is the comparison)."""
    from advmil_amd import ops
    prev = ops.get_gemm_mode()
    try:
        run_case(kind="cluster", lens=(256, 512, 128, 64), gemm_mode=gemm_mode)
    finally:
        ops.set_gemm_mode(prev)


def test_step_without_event_bags_has_no_real_pairs():
    h = run_case(lens=(128, 256, 64), events=(0, 0, 0))


def test_single_bag_step():
    run_case(lens=(512,), events=(1,))


def test_wolabel_mode_partial_visibility():
    # invisible labels: no real pair and no supervised term for those bags (model_handler.py:361-364, 473-480)
    run_case(lens=(128, 256, 64, 512), events=(1, 1, 0, 1), visible=(True, False, True, False), mode="wolabel")


def test_wolabel_mode_nothing_visible():
    run_case(lens=(128, 64), events=(1, 0), visible=(False, False), mode="wolabel")


@pytest.mark.parametrize("which", ["hinge", "wasserstein"])
def test_other_d_losses(which):
    # mean(1+f_fake) + mean(1-f_real) and mean(f_fake) - mean(f_real): the real and fake means nearly cancel in many D
    # gradients (exactly for the logit's additive bias: d/db = 1 - 1 = 0), so Adam's lr*g/|g| updates are round-off noise on
    # BOTH sides; the losses/logits are compared (a few lr = 8e-5 of slack after the D step), the post-step weights are not
    run_case(lens=(128, 256, 64), loss_netD=which, tol=4e-4, check_weights=False)


@pytest.mark.parametrize("which,kind,lens,events,visible", [
    ("hinge", "abmil", (128, 256, 64), None, None), ("wasserstein", "abmil", (128, 256, 64), None, None),
    ("bce", "abmil", (128, 256, 64), None, None),
    # the configuration on which the randomised parity run (tools/probe/oracle_fuzz.py 30 11, case 26) left the 4e-4 band of the
    # post-Adam comparison: DeepAttMISL, one event bag, one invisible label, 4 208 rows (padded slab), wasserstein
    ("wasserstein", "cluster", (128, 896, 3184), (0, 0, 1), (False, True, True)),
    ("hinge", "patch", (1312, 2064, 976), (1, 0, 1), (True, False, True)),
    # round 4, tools/probe/oracle_fuzz.py 40 11 case 25 and 40 23 case 4 (exact fp32 arithmetic, bce): 42 / 75 entries of a first-layer
    # weight (0.03 % / 0.02 %, the post-Adam check of run_case allows 0.01 %) ended one or two sign flips (<= 2 lr per step) away from
    # the oracle after two optimizer steps; the raw gradients of the same configurations agree to 2e-5 of the tensor's scale (here),
    # so the excess is Adam's round-off amplification (step 1 moves the exact-zero-gradient biases by +-lr on both sides, step 2's
    # gradients see it at the 1e-4 level), not a kernel deviation. Logged by the fuzzer with its seeds (profiles/r04_fuzz_counted_cases.jsonl).
    ("bce", "abmil", (4144,), (1,), (False,)), ("bce", "cluster", (592, 3536), (1, 0), None)])
def test_d_loss_gradients_before_adam(which, kind, lens, events, visible):
    gradients_before_adam(which, kind, lens, events, visible)


def gradients_before_adam(which, kind, lens, events, visible, bag_seed0=40, **cfg_over):
    """The hinge / wasserstein D losses (loss/utils.py:182-203) at the contract's tolerance WITHOUT Adam in between: the raw
    gradients of one D backward and one G backward (the arenas the optimizer kernels read) against the oracle's autograd, every
    parameter, 2e-5 of the tensor's gradient scale. (Post-Adam weights of these two losses are round-off noise on both sides:
    their real and fake means nearly cancel, test_other_d_losses -- Adam's first step moves a parameter whose gradient is round-off
    by +-lr with a sign decided by that round-off, and the step's second half, the generator update against the UPDATED
    discriminator, sees it: 2-5e-4 on Loss_G_fake there, 2e-5 here.)"""
    from advmil_amd.model import MyHandler
    nb = len(lens)
    mode = "wlabel" if visible is None else "wolabel"
    h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb, loss_netD=which, **cfg_over), device=DEV)
    dt_ = cfg_over.get("disc_type", "prj")
    PG, PD = load_synth(h.netG, f"G-{kind}:"), load_synth(h.netD, "D-prj:" if dt_ == "prj" else "D-cat:")
    zero_dropout(h.netG); zero_dropout(h.netD)
    bags = []
    for i, n in enumerate(lens):
        y = H.label(i)
        if events is not None:
            y[0, 1] = float(events[i])
        ext = H.T(synth.cluster_ids(0, bag_seed0 + i, n)) if kind == "cluster" else None
        bags.append((H.bag(bag_seed0 + i, max(512, max(lens)))[:, :n].contiguous(), ext, y))
    xs = [[b[0].to(DEV), b[1].to(DEV) if b[1] is not None else torch.zeros(1, 1, device=DEV)] for b in bags]
    ys_host = [b[2] for b in bags]
    ys = [y.to(DEV) for y in ys_host]
    nd = [[H.noise_tensor("gr_d", i, 192)] for i in range(nb)]
    ng = [[H.noise_tensor("gr_g", i, 192)] for i in range(nb)]
    plan = h._plan(xs, ys, mode, None if visible is None else list(visible), ys_host)
    h._disc_backward(0, xs, ys, plan, [[n[0].to(DEV)] for n in nd])
    h._gen_backward(0, xs, ys, plan, [[n[0].to(DEV)] for n in ng])
    torch.cuda.synchronize()
    cfg = O.StepConfig(kind=kind, loss_netD=which, l1_coef=0.0, disc_type=dt_, inner_product=cfg_over.get("disc_prj_iprd", "instance"),
                       prj_path=cfg_over.get("disc_prj_path", "x"))          # (the L1 sub-gradient is applied inside the Adam kernel)
    vis = None if visible is None else list(visible)
    # The oracle in FLOAT64 is the reference here. In fp32 -- the reference's own CPU arithmetic -- the oracle itself is 3.8e-3 (of the
    # tensor's scale) away from float64 on DeepAttMISL's first-layer weight gradient in the `cluster` case below (per-cluster means of
    # 3 184 rows: heavy cancellation); the HIP path is not, and that is what the contract is about.
    def dbl(d):
        return {k_: v_.double() for k_, v_ in d.items()}
    bags64 = [(x_.double(), None if e_ is None else e_.double(), y_.double()) for x_, e_, y_ in bags]
    nd64, ng64 = [[n[0].double()] for n in nd], [[n[0].double()] for n in ng]
    _, gD, _, _ = O.update_disc(cfg, dbl(PG), dbl(PD), bags64, nd64, visible=vis)
    _, gG, _ = O.update_gen(cfg, dbl(PG), dbl(PD), bags64, ng64, visible=vis)
    # Units of a region embedding (FC -> LayerNorm -> ReLU) that hold a LayerNorm output within fp32 round-off of 0 in float64: which ReLU
    # branch such an entry takes in ANY fp32 evaluation is decided by the summation order (a step of 4 000 patches holds 0.5-2 M such outputs:
    # about every second case has one below 3e-7), and the unit's own weight / bias / gamma / beta gradients move by the row's whole
    # contribution when it flips. They are a property of the input, not of a kernel -> not compared; every other
    # entry is.
    Xall = torch.cat([b_[0].reshape(-1, b_[0].shape[-1]) for b_ in bags64])
    skip = {}
    for P_, pre, on in ((PD, "net_pair_one.embedding.", True), (PG, "backbone.patch_embedding_layer.", kind == "patch")):
        if not on or pre + "conv.weight" not in P_:
            continue
        Wd = P_[pre + "conv.weight"].double().reshape(P_[pre + "conv.weight"].shape[0], -1)
        ln = torch.nn.functional.layer_norm(Xall @ Wd.t() + P_[pre + "conv.bias"].double(), (Wd.shape[0],), P_[pre + "norm.weight"].double(),
                                            P_[pre + "norm.bias"].double(), 1e-5)
        units = sorted(set((ln.abs() < 1.5e-6).nonzero()[:, 1].tolist()))
        for suffix in ("conv.weight", "conv.bias", "norm.weight", "norm.bias"):
            skip[pre + suffix] = units
    for net, want in ((h.netD, gD), (h.netG, gG)):
        for k, p in net.named_parameters():
            w = want.get(k)
            got = torch.zeros_like(p) if p.grad is None else p.grad
            if w is None:
                assert float(got.abs().max()) == 0.0, k
                continue
            if skip.get(k):
                keep = torch.ones(w.shape[0], dtype=torch.bool)
                keep[skip[k]] = False
                w, got = w[keep], got[keep.to(got.device)]
            scale = float(w.abs().max())
            # (+ 2.5e-7: gradients that are exactly 0 in exact arithmetic -- the logit's additive bias under hinge / wasserstein, 1 - 1 --
            # come out as fp32 round-off of the cancelling terms here and as 1e-16 in float64)
            assert float((got.cpu().double() - w).abs().max()) <= 2e-5 * scale + 2.5e-7, (which, kind, k, float((got.cpu().double() - w).abs().max()), scale)


@pytest.mark.parametrize("over", [dict(disc_type="cat", disc_prj_path=None), dict(disc_prj_iprd="bag"), dict(disc_prj_path="y"),
                                  dict(disc_prj_iprd="bag", disc_prj_path=None)])
def test_discriminator_variants(over):
    run_case(lens=(128, 256, 64), **over)


def test_checkpoint_save_resume_round_trip(tmp_path):
    from advmil_amd.model import MyHandler
    cfg = default_cfg(bp_every_batch=2, save_path=str(tmp_path))
    xs = [[H.bag(60 + i, 256, DEV), torch.zeros(1, 1, device=DEV)] for i in range(2)]
    ys_host = [H.label(i) for i in range(2)]
    ys = [y.to(DEV) for y in ys_host]
    nz = [[H.noise_tensor("ck", i, 192, DEV)] for i in range(2)]

    def steps(h, n):
        for _ in range(n):
            h._update_disc(0, xs, ys, ys_host=ys_host, noise=nz)
            h._update_gen(0, xs, ys, ys_host=ys_host, noise=nz)

    a = MyHandler(cfg, device=DEV); zero_dropout(a.netG); zero_dropout(a.netD)
    steps(a, 2)
    a.save_model(2, "last", "train")
    ck = torch.load(a._prefixed(a.last_netG_ckpt_path, "train"))
    assert set(ck) == {"epoch", "model", "optimizer"} and set(ck["optimizer"]) == {"state", "param_groups"}
    st0 = ck["optimizer"]["state"][0]
    assert set(st0) >= {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 2.0      # torch.optim.Adam layout
    steps(a, 1)
    b = MyHandler(cfg, device=DEV); zero_dropout(b.netG); zero_dropout(b.netD)
    b.resume_model("last", "train")
    steps(b, 1)
    for pa, pb in ((a.optimizerG, b.optimizerG), (a.optimizerD, b.optimizerD)):
        assert float((pa.flat_param - pb.flat_param).abs().max()) == 0.0
        assert float((pa.flat_m - pb.flat_m).abs().max()) == 0.0 and int(pa.step_t) == int(pb.step_t) == 3


def test_reference_style_adam_state_loads(tmp_path):
    """A torch.optim.Adam state_dict (what the reference's checkpoints hold) loads into FlatAdam."""
    from advmil_amd.model import MyHandler
    h = MyHandler(default_cfg(bp_every_batch=1), device=DEV)
    ref_opt = torch.optim.Adam(h.netD.parameters(), lr=8e-5)
    for p in h.netD.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    ref_opt.step()
    sd = ref_opt.state_dict()
    h.optimizerD.load_state_dict(sd)
    assert int(h.optimizerD.step_t) == 1
    p0 = next(iter(h.netD.parameters()))
    assert torch.allclose(h.optimizerD.state[p0]["exp_avg"], ref_opt.state[p0]["exp_avg"])


def test_graph_recapture_after_lr_change():
    from advmil_amd.graphed import GraphedStep
    from advmil_amd.model import MyHandler
    h = MyHandler(default_cfg(bp_every_batch=2), device=DEV)
    xs = [[H.bag(70 + i, 256, DEV), torch.zeros(1, 1, device=DEV)] for i in range(2)]
    ys_host = [H.label(i) for i in range(2)]
    g = GraphedStep(h, xs, [y.to(DEV) for y in ys_host], ys_host, warmup=1)
    g.replay()
    for grp in h.optimizerG.param_groups:
        grp["lr"] *= 0.5                                  # what ReduceLROnPlateau does (model_handler.py:109)
    g.replay()                                            # must re-capture with the new launch constant
    assert g.lrs[0] == h.optimizerG.param_groups[0]["lr"]
    torch.cuda.synchronize()
    assert torch.isfinite(h.optimizerG.flat_param).all()


def test_ingest_stager_makes_the_step_slab_zero_copy():
    """CPU bags -> pinned slab -> device slab on the copy stream: the step's [sum N, 1024] matrix is a view of the staging
    buffer (no concatenation), two consecutive steps use the two buffer pairs, and results equal the oracle (checked by
    every other test in this file, which all go through the same path)."""
    from advmil_amd.model import MyHandler
    h = MyHandler(default_cfg(bp_every_batch=3), device=DEV)
    lens = (64, 256, 128)
    loader = [(torch.tensor([[i]], dtype=torch.int), [H.bag(80 + i, 512)[:, :lens[i % 3]].contiguous(), torch.zeros(1, 1)], H.label(i))
              for i in range(6)]
    seen = []
    orig = h._slab
    h._slab = lambda xs, plan=None: seen.append(orig(xs, plan)) or seen[-1]
    h._train_each_epoch(loader, "train")
    st = h._stager
    bases = {st.dev[0].data_ptr(), st.dev[1].data_ptr()}
    assert len(seen) == 4 and {t.data_ptr() for t in seen} == bases          # D and G phase of two steps, two buffer pairs
    assert seen[0] is seen[1] and seen[2] is seen[3]                         # one slab per step plan, shared by the D and the G update
    assert all(tuple(t.shape) == (sum(lens), 1024) for t in seen)
    torch.cuda.synchronize()
    want = torch.cat([loader[3 + j][1][0][0] for j in range(3)], dim=0)       # second step's bags, in order
    assert torch.equal(seen[-1].cpu(), want)


@pytest.mark.parametrize("nrows", [64, 2048])
def test_second_epoch_is_served_from_the_device_resident_bag_cache(nrows):
    """Epoch 1 stages the host bags over PCIe and keeps each one (with its bf16x3 operand planes when the slab is large enough
    to use them) in HBM, keyed by the loader's patient index; epoch 2 over a SHUFFLED loader touches no host bag (the host
    tensors are poisoned in between) and equals, bit for bit, a handler that stages every epoch (ADVMIL_BAG_CACHE_GB=0);
    an LRU budget that holds only part of the cohort still gives the same numbers. Replaces the per-epoch `.cuda()` of
    reference model/model_handler.py:315."""
    from advmil_amd.model import MyHandler
    lens = (nrows, 2 * nrows, nrows // 2 * 3)

    def mk():
        return [(torch.tensor([[i]], dtype=torch.int), [H.bag(80 + i, 3 * nrows)[:, :lens[i % 3]].contiguous(), torch.zeros(1, 1)], H.label(i))
                for i in range(6)]

    class DS:                                    # a dataset object: the cache's scope (a bare list is only cached on request)
        def __init__(self, items):
            self.items = items

    class DL:
        def __init__(self, ds, order):
            self.dataset, self.order = ds, order

        def __iter__(self):
            return iter([self.dataset.items[i] for i in self.order])

    def run(cache_gb, poison):
        from advmil_amd import ingest
        ingest.device_bag_cache(DEV).clear()     # (the device's cache is shared: bags of earlier tests would hold the small budget)
        h = MyHandler(default_cfg(bp_every_batch=3, bag_cache_gb=cache_gb), device=DEV)
        load_synth(h.netG, "G-abmil:"); load_synth(h.netD, "D-prj:")
        h.rng.reset(99)
        ds = DS(mk())
        h._train_each_epoch(DL(ds, list(range(6))), "train")
        order = [4, 0, 5, 2, 1, 3]                                             # epoch 2: shuffled, step batches regrouped
        if poison:
            torch.cuda.synchronize()
            ds.items = [(it[0], [H.poison_host_bag(it[1][0]), it[1][1]], it[2]) for it in ds.items]
        cl = h._train_each_epoch(DL(ds, order), "train")
        view = h._bag_caches.get("train")      # (this loader's window onto the device's ONE shared cache: stats taken now)
        return cl, h.pop_logs(), h.optimizerG.flat_param.clone(), None if view is None else view.stats()

    a = run(None, True)                  # default budget: everything resident, the poisoned host bags are never read
    b = run(0, False)                    # no cache: every epoch over PCIe
    one = (lens[0] + lens[1]) * 1024 * 8 / 1e9
    c = run(one, False)                  # a budget of about two bags: the rest is refused, mixed cached / staged step batches
    assert a[3]["bags"] == 6 and a[3]["hits"] == 6 and b[3] is None and c[3]["refused"] > 0 and 0 < c[3]["hits"] < 6
    for other in (b, c):
        assert torch.equal(a[0]["y_hat"], other[0]["y_hat"]) and torch.equal(a[0]["f_fake"], other[0]["f_fake"])
        assert torch.equal(a[2], other[2])
        for la, lb in zip(a[1], other[1]):
            for k in la:
                assert float(la[k]) == float(lb[k]), k


@pytest.mark.parametrize("kind", ["abmil", "patch", "cluster"])
def test_forward_memo_is_bitwise_neutral(kind):
    """The G-step reuses the row-sized pre-dropout layer outputs of the D-step's eval forward (ops.ForwardMemo). With
    dropout ON, two optimizer steps with the memo must equal two steps without it bit for bit."""
    from advmil_amd import ops
    from advmil_amd.model import MyHandler

    def run(min_rows):
        old = ops.MEMO_MIN_ROWS
        ops.MEMO_MIN_ROWS = min_rows
        try:
            nb, lens = 3, (256, 128, 64)
            h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb), device=DEV)
            load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
            h.rng.reset(1234)
            loader = []
            for i in range(2 * nb):
                x = H.bag(70 + i, 512)[:, :lens[i % nb]].contiguous()
                ext = H.T(synth.cluster_ids(0, i, lens[i % nb])) if kind == "cluster" else torch.zeros(1, 1)
                loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext], H.label(i)))
            cl = h._train_each_epoch(loader, "train", "wlabel")
            logs = h.pop_logs()
            return cl, logs, {k: v.clone() for k, v in h.netG.state_dict().items()}, {k: v.clone() for k, v in h.netD.state_dict().items()}
        finally:
            ops.MEMO_MIN_ROWS = old

    a, b = run(1), run(10 ** 9)
    assert torch.equal(a[0]["y_hat"], b[0]["y_hat"]) and torch.equal(a[0]["f_fake"], b[0]["f_fake"])
    for la, lb in zip(a[1], b[1]):
        for k in la:
            assert float(la[k]) == float(lb[k]), k
    for sa, sb in ((a[2], b[2]), (a[3], b[3])):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("kind", ["abmil", "patch"])
def test_two_layer_launch_is_bitwise_neutral(kind):
    """The D update runs the generator's and the discriminator's first layer over the step slab as ONE plane-fed launch
    (ops.prefill_two_layers: X staged once). Same products in the same order: two optimizer steps with it must equal two steps with
    the layers launched separately, bit for bit (dropout ON, bf16x3 arithmetic, slab-sized bags)."""
    from advmil_amd import ops
    from advmil_amd.model import MyHandler

    def run(two):
        old, old_h = ops.TWO_LAYERS, ops.H_PLANES_ONLY
        ops.TWO_LAYERS = two
        # (the launch itself is what this test pins: with the round-6 planes-only output of layer 1 the two paths differ by design -- the
        # pooling then reads hi + lo instead of fp32 rows; tests/test_h_planes_gpu.py covers that mode)
        ops.H_PLANES_ONLY = False
        try:
            nb, n = 8, 16384                                  # 131072 rows: the size class where the slab keeps resident planes
            calls = []
            real = ops.gemm_two_layers
            ops.gemm_two_layers = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb, gemm_mode="bf16x3"), device=DEV)
            load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
            h.optimizerG.refresh_planes(); h.optimizerD.refresh_planes()
            h.rng.reset(99)
            X = torch.randn(nb * n, 1024, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
            xs = [[X[i * n:(i + 1) * n].unsqueeze(0), torch.zeros(1, 1)] for i in range(nb)]
            ys = [H.label(i).to(DEV) for i in range(nb)]
            ys_host = [H.label(i) for i in range(nb)]
            for i in range(2):
                plan = h._plan(xs, ys, "wlabel", None, ys_host)
                h._update_disc(i, xs, ys, "wlabel", None, ys_host=ys_host, plan=plan)
                h._update_gen(i, xs, ys, "wlabel", None, ys_host=ys_host, plan=plan)
                h.rng.advance(1)
            torch.cuda.synchronize()
            assert len(calls) == (2 if two else 0), calls      # one fused launch per D update -- or none
            return h.pop_logs(), {k: v.clone() for k, v in h.netG.state_dict().items()}, {k: v.clone() for k, v in h.netD.state_dict().items()}
        finally:
            ops.TWO_LAYERS, ops.H_PLANES_ONLY = old, old_h
            ops.gemm_two_layers = real
            ops.set_gemm_mode("exact")

    a, b = run(True), run(False)
    for la, lb in zip(a[0], b[0]):
        for k in la:
            assert float(la[k]) == float(lb[k]), k
    for sa, sb in ((a[1], b[1]), (a[2], b[2])):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


def test_forward_memo_is_consumed_and_cleared():
    from advmil_amd import ops
    from advmil_amd.model import MyHandler
    old = ops.MEMO_MIN_ROWS
    ops.MEMO_MIN_ROWS = 1
    try:
        h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=2), device=DEV)
        xs = [[H.bag(90 + i, 256).to(DEV), None] for i in range(2)]
        ys = [H.label(i).to(DEV) for i in range(2)]
        plan = h._plan(xs, ys, "wlabel", None, None)
        h._update_disc(1, xs, ys, plan=plan)
        assert len(ops.MEMO.store) == 1 and ops.MEMO.mode is None          # the FC output of the eval forward
        h._update_gen(1, xs, ys, plan=plan)                                 # same step plan: replayed and consumed
        assert len(ops.MEMO.store) == 0
        plan = h._plan(xs, ys, "wlabel", None, None)
        h._update_disc(2, xs, ys, plan=plan)
        h.optimizerG.step()                                                 # weights moved: the stale entry must not be used
        h._update_gen(2, xs, ys, plan=plan)
        assert len(ops.MEMO.store) == 0
        # unpaired calls (each builds its own plan -> its own token): the eval forward's entry is never replayed into another
        # step's train forward, even though slab address, shape and weights coincide
        h._update_disc(3, xs, ys)
        assert len(ops.MEMO.store) == 1
        h._update_gen(3, xs, ys)
        assert len(ops.MEMO.store) == 0
    finally:
        ops.MEMO_MIN_ROWS = old


def test_two_generator_updates_per_step():
    """gen_updates = 2 (model_handler.py:341-342): the second generator update sees updated G weights, so the forward memo of the
    eval pass must not be reused for it, and D stays frozen in both; compared with the oracle's update_gen applied twice."""
    from advmil_amd.model import MyHandler
    nb, lens = 3, (256, 128, 64)
    h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=nb, gen_updates=2), device=DEV)
    PG, PD = load_synth(h.netG, "G-abmil:"), load_synth(h.netD, "D-prj:")
    zero_dropout(h.netG); zero_dropout(h.netD)
    bags, loader = [], []
    for i in range(nb):
        x = H.bag(60 + i, 512)[:, :lens[i]].contiguous()
        y = H.label(i)
        bags.append((x, None, y)); loader.append((torch.tensor([[i]], dtype=torch.int), [x, torch.zeros(1, 1)], y))
    nd = [[H.noise_tensor("gu_d", i, 192)] for i in range(nb)]
    ng = [[H.noise_tensor("gu_g", i, 192)] for i in range(nb)]
    h.noise_hook = lambda ph, i: [(nd if ph == "d" else ng)[i][0].to(DEV)]
    from advmil_amd import ops
    old = ops.MEMO_MIN_ROWS
    ops.MEMO_MIN_ROWS = 1                                   # make the memo active at this size
    try:
        h._train_each_epoch(loader, "train", "wlabel")
    finally:
        ops.MEMO_MIN_ROWS = old
    logs = h.pop_logs()
    assert len(logs) == 3                                   # D, G, G
    cfg = O.StepConfig(kind="abmil")
    stG, stD = {}, {}
    ld, gD, _, _ = O.update_disc(cfg, PG, PD, bags, nd, None, None)
    PD2 = O.adam_step(PD, gD, stD, cfg.lr_d, 0.0, decay_filter=False)
    oPG = PG
    for k in range(2):
        lg, gG, _ = O.update_gen(cfg, oPG, PD2, bags, ng, None)
        oPG = O.adam_step(oPG, gG, stG, cfg.lr_g, cfg.wd_g, decay_filter=True)
        got = {kk.split("/")[-1]: v for kk, v in logs[1 + k].items()}
        for kk in ("Loss_G_fake", "Loss_G_time", "Loss_G_total"):
            assert abs(got[kk] - lg[kk]) < 2e-5, (k, kk, got[kk], lg[kk])
    for kk, v in h.netD.state_dict().items():               # D only moved by its own update
        if kk.endswith("pool.fc2.bias"):
            continue
        assert float((v.cpu() - PD2[kk]).abs().max()) < 5e-5, kk


@pytest.mark.parametrize("kind", ["abmil", "patch", "cluster"])
def test_slab_pad_is_numerically_neutral(kind, monkeypatch):
    """The rows of a real step batch are a multiple of 16, not of the slab kernels' 256-row tiles: the staging slab appends zero rows
    (a dummy bag, SlabStager.pad_rows) so that the fast forms apply. Two optimizer steps with the pad (default) against two steps
    without it (ADVMIL_SLAB_PAD=0): predictions, D scores, logged losses and updated weights agree to round-off -- the pad rows
    contribute exact zeros to every gradient -- and the padded run really took whole tiles."""
    from advmil_amd.model import MyHandler
    lens = (2048, 1040, 1536, 912)                         # 5536 rows per step: 5536 % 256 = 160 -> 96 pad rows

    def run(pad):
        monkeypatch.setenv("ADVMIL_SLAB_PAD", str(pad))
        h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=4, bag_cache_gb=0), device=DEV)
        load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
        h.rng.reset(77)
        seen = []
        orig = h._slab_build
        h._slab_build = lambda xs, p=0: seen.append(int(orig(xs, p).shape[0])) or orig(xs, p)
        loader = []
        for i in range(8):
            n = lens[i % 4]
            ext = H.T(synth.cluster_ids(0, 600 + i, n)) if kind == "cluster" else torch.zeros(1, 1)
            loader.append((torch.tensor([[i]], dtype=torch.int), [H.bag(600 + i, 2048)[:, :n].contiguous(), ext], H.label(i)))
        cl = h._train_each_epoch(loader, "train")
        return cl, h.pop_logs(), h.optimizerG.flat_param.clone(), h.optimizerD.flat_param.clone(), seen

    a, b = run(256), run(0)
    assert set(a[4]) == {5632} and set(b[4]) == {5536}
    for k in ("y_hat", "f_fake"):
        assert float((a[0][k] - b[0][k]).abs().max()) <= 2e-6, k
    for la, lb in zip(a[1], b[1]):
        for k in la:
            assert abs(float(la[k]) - float(lb[k])) <= 2e-6 * max(1.0, abs(float(lb[k]))), k
    for i in (2, 3):                                       # Adam turns round-off in g ~ 0 into 2 lr at most; the bulk must agree
        d = (a[i] - b[i]).abs()
        assert float(d.max()) <= 2.5e-4 and float((d > 1e-6).float().mean()) < 0.02, (i, float(d.max()), float((d > 1e-6).float().mean()))


@pytest.mark.parametrize("kind", ["abmil", "patch"])
def test_x_storage_bf16_equals_fp32_storage_on_rounded_bags(kind):
    """cfg x_storage = 'bf16' (bags held as ONE bf16 plane: staging slab, device-resident cache; two MFMAs per product on the slab
    contractions) is the fp32-storage step on the bf16-rounded bags, bit for bit: the products it drops are exact zeros. Checked
    through the product loop with fp32 host bags (rounded on the copy stream behind their H2D copy), with bf16 host bags (DMA'd
    as they are: half the PCIe bytes), and on a second epoch served from the bf16 cache."""
    from advmil_amd import ingest, ops
    from advmil_amd.model import MyHandler
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        lens = (2048, 1024, 3072, 2048)

        class DS:
            def __init__(self, items):
                self.items = items

        class DL:
            def __init__(self, ds):
                self.dataset = ds

            def __iter__(self):
                return iter(self.dataset.items)

        def run(x_storage, host):
            ingest.device_bag_cache(DEV).clear()
            h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=4, x_storage=x_storage), device=DEV)
            load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
            h.rng.reset(31)
            items = []
            for i, n in enumerate(lens):
                x = H.bag(700 + i, 3072)[:, :n].contiguous()
                xb = x.to(torch.bfloat16)
                xh = {"fp32": x, "rounded": xb.float(), "bf16": xb}[host]
                items.append((torch.tensor([[i]], dtype=torch.int), [xh, torch.zeros(1, 1)], H.label(i)))
            dl = DL(DS(items))
            out = []
            for _ in range(2):
                cl = h._train_each_epoch(dl, "train")
                out.append((cl["y_hat"].clone(), cl["f_fake"].clone()))
            view = h._bag_caches["train"]
            return out, h.optimizerG.flat_param.clone(), None if view is None else view.stats(), h

        ref, wref, _, _ = run("fp32", "rounded")
        for host in ("fp32", "bf16"):
            got, w, st, h = run("bf16", host)
            assert st["bags"] == 4 and st["hits"] == 4 and st["gb"] < 4 * 3072 * 1024 * 2 / 1e9 + 1e-6       # 2 bytes per element
            assert h._stager.store == torch.bfloat16
            for (ya, fa), (yb, fb) in zip(ref, got):
                assert torch.equal(ya, yb) and torch.equal(fa, fb), host
            assert torch.equal(wref, w), host
    finally:
        ops.set_gemm_mode(prev)


def test_cached_bags_enter_the_step_slab_as_operand_planes_only():
    """bf16x3: a bag served from the device-resident cache is staged WITHOUT its fp32 rows (ingest.SlabStager.add_device: the launch derives
    the operand planes straight from the cache entry), the step slab is flagged and every contraction reads planes only. Same numbers,
    bit for bit, as with the rows copied (ADVMIL_STAGE_PLANES_ONLY=0) -- with the slab's stale fp32 rows poisoned in between, so that any
    kernel still reading them would show."""
    from advmil_amd import ingest
    from advmil_amd.model import MyHandler
    lens = (2048, 4096, 3056)            # (a step batch of 9200 rows: 16 zero rows of slab pad behind the bags)

    def mk():
        return [(torch.tensor([[i]], dtype=torch.int), [H.bag(60 + i, 4096)[:, :lens[i % 3]].contiguous(), torch.zeros(1, 1)], H.label(i))
                for i in range(6)]

    class DS:
        def __init__(self, items):
            self.items = items

    class DL:
        def __init__(self, ds, order):
            self.dataset, self.order = ds, order

        def __iter__(self):
            return iter([self.dataset.items[i] for i in self.order])

    def run(planes_only):
        from advmil_amd import ops
        old = ingest.PLANES_ONLY_STAGE
        prev_mode = ops.get_gemm_mode()
        ingest.PLANES_ONLY_STAGE = planes_only
        seen = []
        orig = MyHandler._slab_build_static

        def spy(xs, resident_planes=True, pad=0):
            X = orig(xs, resident_planes, pad)
            st = bool(getattr(X, "_advmil_fp32_stale", False))
            seen.append(st)
            if st:
                X.fill_(float("nan"))         # nothing may read these rows (the planes are what the step computes on)
            return X
        MyHandler._slab_build_static = staticmethod(spy)
        try:
            ingest.device_bag_cache(DEV).clear()
            h = MyHandler(default_cfg(bp_every_batch=3, gemm_mode="bf16x3"), device=DEV)
            load_synth(h.netG, "G-abmil:"); load_synth(h.netD, "D-prj:")
            h.rng.reset(5)
            ds = DS(mk())
            h._train_each_epoch(DL(ds, list(range(6))), "train")
            n1 = len(seen)
            cl = h._train_each_epoch(DL(ds, [4, 0, 5, 2, 1, 3]), "train")
            torch.cuda.synchronize()
            return cl, h.pop_logs(), h.optimizerG.flat_param.clone(), h.optimizerD.flat_param.clone(), seen[:n1], seen[n1:]
        finally:
            MyHandler._slab_build_static = staticmethod(orig)
            ingest.PLANES_ONLY_STAGE = old
            ops.set_gemm_mode(prev_mode)

    a, b = run(True), run(False)
    assert not any(a[4]) and all(a[5]) and len(a[5]) > 0          # epoch 1 stages host bags (rows valid), epoch 2 planes only
    assert not any(b[4]) and not any(b[5])
    assert torch.equal(a[0]["y_hat"], b[0]["y_hat"]) and torch.equal(a[0]["f_fake"], b[0]["f_fake"])
    assert torch.isfinite(a[2]).all() and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for la, lb in zip(a[1], b[1]):
        for k in la:
            assert float(la[k]) == float(lb[k]), k
