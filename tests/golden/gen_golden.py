#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference
(/root/reference, liupei101/AdvMIL @ v1) on CPU through import shims, and pin the oracle
(oracle/advmil_oracle.py) against it in the same run.

Runs only in the build container (the reference never travels). Inputs and weights are
regenerated from the repo's counter RNG (advmil_amd/synth.py), so the fixtures hold only
outputs (KB-sized). Usage:  python tests/golden/gen_golden.py

Shims: stub modules for wandb / h5py / torch_geometric /
torch_sparse, a ReduceLROnPlateau that swallows `verbose` (model_handler.py:109 on
torch 2.10), `.cuda()` no-ops, cwd=/root/reference.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from advmil_amd import synth  # noqa: E402
from oracle import advmil_oracle as O  # noqa: E402

DATA_SEED, PARAM_SEED = 0, 42
LOG = []


# ---------------------------------------------------------------------------------------
def install_shims():
    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    class _D:
        def __init__(self, *a, **k):
            pass

    stub("wandb", init=lambda **k: None, log=lambda d, **k: LOG.append(dict(d)), Image=lambda x: x)
    stub("h5py")
    tg = stub("torch_geometric", is_debug_enabled=lambda: False)
    tg.nn = stub("torch_geometric.nn", GENConv=_D, DeepGCNLayer=_D)
    tg.data = stub("torch_geometric.data", Data=_D, Batch=_D)
    stub("torch_sparse", SparseTensor=_D, cat=lambda *a, **k: None)
    from torch.optim import lr_scheduler as ls
    _R = ls.ReduceLROnPlateau

    class _RP(_R):
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)

    ls.ReduceLROnPlateau = _RP
    torch.cuda.set_device = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    os.chdir(REF)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_synth(module, seed=PARAM_SEED, prefix=""):
    sd = module.state_dict()
    new = {k: T(synth.param(seed, prefix + k, tuple(v.shape))) for k, v in sd.items()}
    module.load_state_dict(new)
    return {k: v.clone() for k, v in new.items()}


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


class NoiseQueue:
    """Replaces utils.func.generate_noise (func.py:154-164) with pre-generated tensors."""

    def __init__(self):
        self.q = []

    def __call__(self, *dims, to_device="cpu", distribution="uniform"):
        n = self.q.pop(0)
        assert tuple(n.shape) == tuple(dims), (n.shape, dims)
        return n


def noise_tensor(tag, k, width):
    return T(synth.device_uniform(DATA_SEED, synth.stream_key(7, f"{tag}:{k}"), width).reshape(1, width))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def a_summary(A):
    A = A.reshape(-1).double()
    ent = float(-(A * torch.log(A.clamp_min(1e-300))).sum())
    return {"sum": float(A.sum()), "max": float(A.max()), "argmax": int(A.argmax()), "entropy": ent}


# ---------------------------------------------------------------------------------------
def build_generator(kind):
    from types import SimpleNamespace
    from model.backbone import load_backbone
    from model.GANSurv import Generator
    bb = load_backbone(kind, [1024, 384, 384])
    g = Generator(384, 1, bb, SimpleNamespace(noise=[0, 1], hops=1, noise_dist="uniform"), False, 0.6, "sigmoid")
    return g


def build_disc(disc_type="prj", iprd="instance", prj_path="x"):
    from types import SimpleNamespace
    from model.GANSurv import Discriminator, PrjDiscriminator
    ax = SimpleNamespace(in_dim=1024, out_dim=128, ksize=1, backbone="avgpool", dropout=0.25)
    ay = SimpleNamespace(in_dim=1, hid_dims=[64, 128], norm=False, dropout=0.0)
    if disc_type == "prj":
        return PrjDiscriminator(ax, ay, prj_path=prj_path, inner_product=iprd)
    return Discriminator(ax, ay)


def gen_G1(out, pin):
    """Eval mode, zero_noise: attention A, H, y per backbone x N."""
    for kind in ("abmil", "patch", "cluster"):
        g = build_generator(kind).eval()
        P = load_synth(g, prefix=f"G-{kind}:")
        for N in (512, 1024, 8192):
            x = T(synth.bag(DATA_SEED, 0, N))
            ext = T(synth.cluster_ids(DATA_SEED, 0, N)) if kind == "cluster" else None
            cap = {}
            if kind == "patch":
                hk = g.backbone.pool.fc2.register_forward_hook(lambda m, i, o: cap.__setitem__("s", o.detach()))
            else:
                hk = g.backbone.attention_net[3].register_forward_hook(lambda m, i, o: cap.__setitem__("s", o[0].detach()))
            cap_h = {}
            hk2 = g.backbone.register_forward_hook(lambda m, i, o: cap_h.__setitem__("H", o.detach()))
            with torch.no_grad():
                y = g(x, ext, zero_noise=True)
            hk.remove(); hk2.remove()
            s = cap["s"].reshape(-1)
            A = torch.softmax(s, dim=0)
            yo, Ao, Ho = O.generator(P, x, ext, kind, (0, 1), None, None, "sigmoid", return_attn=True)
            pin[f"G1/{kind}/{N}"] = {"y": maxdiff(y, yo), "A": maxdiff(A, Ao.reshape(-1)), "H": maxdiff(cap_h["H"], Ho)}
            key = f"G1_{kind}_{N}"
            out[key + "_y"] = y.numpy()
            out[key + "_H"] = cap_h["H"].numpy()
            if N <= 1024:
                out[key + "_A"] = A.numpy()
            else:
                out[key + "_A_strided"] = A[::32].numpy()
            out[key + "_Astat"] = np.array(list(a_summary(A).values()), dtype=np.float64)


def gen_G2(out, pin):
    """MyHandler.test_model (model_handler.py:598-643): y_hat, f_fake, 30 samples + median."""
    import utils.func
    import model.GANSurv as GS
    from model.model_handler import MyHandler
    nq = NoiseQueue()
    old = (utils.func.generate_noise, GS.generate_noise)
    utils.func.generate_noise = nq
    GS.generate_noise = nq
    try:
        for kind in ("abmil", "patch"):
            g = build_generator(kind)
            d = build_disc()
            PG = load_synth(g, prefix=f"G-{kind}:")
            PD = load_synth(d, prefix="D-prj:")
            N, nb, K = 512, 2, 30
            loader, noises = [], []
            for i in range(nb):
                x = T(synth.bag(DATA_SEED, i, N))
                ext = torch.zeros(1, 1)
                y = T(synth.label(DATA_SEED, i))
                loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext], y))
                ns = [noise_tensor(f"G2:{kind}:{i}", k, 192) for k in range(K + 1)]
                noises.append(ns)
                nq.q.extend(ns)
            res = MyHandler.test_model(g, d, kind, loader, times_test_sample=K, checkpoints=None, test_zero_noise=False)
            cfg = O.StepConfig(kind=kind)
            for i in range(nb):
                yh, ff, dist, avg = O.test_model_bag(cfg, PG, PD, loader[i][1][0], None,
                                                     [noises[i][0]], [[n] for n in noises[i][1:]])
                pin[f"G2/{kind}/bag{i}"] = {
                    "y_hat": maxdiff(res["y_hat"][i], yh.reshape(-1)), "f_fake": maxdiff(res["f_fake"][i], ff.reshape(-1)),
                    "dist": maxdiff(res["dist_y_hat"][i], dist.reshape(K, 1)), "avg": maxdiff(res["avg_y_hat"][i], avg.reshape(-1))}
            for k in ("y_hat", "f_fake", "dist_y_hat", "avg_y_hat"):
                out[f"G2_{kind}_{k}"] = res[k].numpy()
    finally:
        utils.func.generate_noise, GS.generate_noise = old


def gen_G3(out, pin):
    """Discriminator variants in eval mode."""
    N = 512
    x = T(synth.bag(DATA_SEED, 3, N))
    t = torch.tensor([[0.37]])
    for disc_type, iprd, prj in (("prj", "instance", "x"), ("prj", "bag", "x"), ("prj", "instance", "y"),
                                 ("prj", "bag", None), ("cat", "bag", None)):
        d = build_disc(disc_type, iprd, prj).eval()
        name = f"D-{disc_type}-{iprd}-{prj}"
        P = load_synth(d, prefix="D-prj:" if disc_type == "prj" else "D-cat:")
        with torch.no_grad():
            f = d(x, t)
            hid_x, fc_ins = d.net_pair_one(x, return_instance=True)
            hid_t = d.net_pair_two(t)
        if disc_type == "prj":
            fo = O.prj_discriminator(P, x, t, iprd, prj, None)
        else:
            fo = O.discriminator_cat(P, x, t, None)
        hxo, fio, _ = O.embed_x_layer(O._sub(P, "net_pair_one."), x, None)
        pin[f"G3/{name}"] = {"f": maxdiff(f, fo), "hid_x": maxdiff(hid_x, hxo), "fc_ins": maxdiff(fc_ins, fio)}
        out[f"G3_{name}_f"] = f.numpy()
        out[f"G3_{name}_hid_x"] = hid_x.numpy()
        out[f"G3_{name}_hid_t"] = hid_t.numpy()
        out[f"G3_{name}_fc_ins_mean"] = fc_ins.mean(dim=1).numpy()


def tensor_stats(sd):
    return {k: [float(v.double().sum()), float(v.double().norm())] for k, v in sd.items()}


def gen_G4(out, pin):
    """Two full optimizer steps through the REAL MyHandler._train_each_epoch
    (model_handler.py:301-347), dropout p=0, injected noise, 16 bags (8 events) per step."""
    import utils.func
    import model.GANSurv as GS
    from model.model_handler import MyHandler
    cfg0 = yaml.load(open(os.path.join(REF, "config/cfg_nlst.yaml")), Loader=yaml.FullLoader)
    nq = NoiseQueue()
    old = (utils.func.generate_noise, GS.generate_noise)
    utils.func.generate_noise = nq
    GS.generate_noise = nq
    try:
        for kind, N in (("abmil", 512), ("patch", 512)):
            cfg = dict(cfg0)
            cfg.update(bcb_mode=kind, data_split_seed=0, save_path=f"/tmp/advmil_golden_{kind}", wandb_dir="/tmp",
                       num_workers=0, bp_every_batch=16)
            h = MyHandler(cfg)
            PG = load_synth(h.netG, prefix=f"G-{kind}:")
            PD = load_synth(h.netD, prefix="D-prj:")
            zero_dropout(h.netG); zero_dropout(h.netD)
            nb = 32
            h.patient_id["label_visible"] = h.patient_id["train"] = [str(i) for i in range(nb)]
            loader, bags = [], []
            for i in range(nb):
                x = T(synth.bag(DATA_SEED, i, N)); y = T(synth.label(DATA_SEED, i))
                loader.append((torch.tensor([[i]], dtype=torch.int), [x, torch.zeros(1, 1)], y))
                bags.append((x, None, y))
            noise_d = [[noise_tensor(f"G4d:{kind}", i, 192)] for i in range(nb)]
            noise_g = [[noise_tensor(f"G4g:{kind}", i, 192)] for i in range(nb)]
            for s in range(2):
                nq.q.extend([n[0] for n in noise_d[16 * s:16 * s + 16]])
                nq.q.extend([n[0] for n in noise_g[16 * s:16 * s + 16]])
            LOG.clear()
            cl = h._train_each_epoch(loader, "train")
            logs = [{k.split("/")[-1]: v for k, v in d.items()} for d in LOG]
            # oracle replay of the same two steps
            ocfg = O.StepConfig(kind=kind)
            stG, stD = {}, {}
            oPG, oPD = PG, PD
            ologs, oy, of = [], [], []
            for s in range(2):
                sl = slice(16 * s, 16 * s + 16)
                oPG, oPD, lg, yh, ff, gG, gD = O.train_step(ocfg, oPG, oPD, stG, stD, bags[sl], noise_d[sl], noise_g[sl])
                ologs.append(lg); oy.append(yh); of.append(ff)
                if s == 0:
                    g1G, g1D = gG, gD
            refG = {k: v.detach() for k, v in h.netG.state_dict().items()}
            refD = {k: v.detach() for k, v in h.netD.state_dict().items()}
            pin[f"G4/{kind}"] = {
                "post_G": max(maxdiff(refG[k], oPG[k]) for k in refG),
                "post_D": max(maxdiff(refD[k], oPD[k]) for k in refD),
                "y_hat": maxdiff(cl["y_hat"].reshape(-1), torch.cat(oy).reshape(-1)),
                "f_fake": maxdiff(cl["f_fake"].reshape(-1), torch.cat(of).reshape(-1)),
                "logs": max(abs(logs[2 * s + j][k] - ologs[s][k]) for s in range(2) for j, ks in
                            ((0, ("Loss_D", "D_real", "D_fake")), (1, ("Loss_G_fake", "Loss_G_time", "Loss_G_total")))
                            for k in ks),
                # second-step G grads still sit in .grad after the epoch
                "grad_G_step2": max(maxdiff(p.grad, gG[k]) for k, p in h.netG.named_parameters()),
            }
            out[f"G4_{kind}_logs"] = np.array(
                [[logs[2 * s][k] for k in ("Loss_D", "D_real", "D_fake")] +
                 [logs[2 * s + 1][k] for k in ("Loss_G_fake", "Loss_G_time", "Loss_G_total", "D_fake_avg")]
                 for s in range(2)], dtype=np.float64)
            out[f"G4_{kind}_y_hat"] = cl["y_hat"].numpy()
            out[f"G4_{kind}_f_fake"] = cl["f_fake"].numpy()
            out[f"G4_{kind}_y"] = cl["y"].numpy()
            keysG = sorted(refG); keysD = sorted(refD)
            out[f"G4_{kind}_keysG"] = np.array(keysG); out[f"G4_{kind}_keysD"] = np.array(keysD)
            out[f"G4_{kind}_postG_stats"] = np.array([tensor_stats(refG)[k] for k in keysG])
            out[f"G4_{kind}_postD_stats"] = np.array([tensor_stats(refD)[k] for k in keysD])
            # parameter deltas after 2 Adam steps (sum, L2) -- far more sensitive than the raw stats
            out[f"G4_{kind}_dG_stats"] = np.array([[float((refG[k].double() - PG[k].double()).sum()),
                                                    float((refG[k].double() - PG[k].double()).norm())] for k in keysG])
            out[f"G4_{kind}_dD_stats"] = np.array([[float((refD[k].double() - PD[k].double()).sum()),
                                                    float((refD[k].double() - PD[k].double()).norm())] for k in keysD])
            gk = [k for k, _ in h.netG.named_parameters()]
            out[f"G4_{kind}_gradG2_keys"] = np.array(gk)
            out[f"G4_{kind}_gradG2_norm"] = np.array([float(p.grad.double().norm()) for _, p in h.netG.named_parameters()])
            out[f"G4_{kind}_gradG1_norm_oracle"] = np.array([float(g1G[k].double().norm()) for k in gk])
            dk = [k for k, _ in h.netD.named_parameters()]
            out[f"G4_{kind}_gradD1_keys"] = np.array(dk)
            out[f"G4_{kind}_gradD1_norm_oracle"] = np.array([float(g1D[k].double().norm()) if k in g1D else 0.0 for k in dk])
    finally:
        utils.func.generate_noise, GS.generate_noise = old


def gen_G5(out, pin):
    """Loss functions alone (loss/utils.py) on random vectors, incl. real=None."""
    from loss.utils import real_fake_loss, fake_generator_loss, recon_loss, loss_reg_l1
    real = T(synth.normal(synth.stream_key(1, "G5real"), 7))
    fake = T(synth.normal(synth.stream_key(1, "G5fake"), 16))
    p = T(synth.uniform01(synth.stream_key(1, "G5p"), 16)); t = T(synth.uniform01(synth.stream_key(1, "G5t"), 16))
    e = (torch.arange(16) % 2).float()
    vals, d = [], 0.0
    for which in ("bce", "hinge", "wasserstein"):
        for r in (real, None):
            a = real_fake_loss(None if r is None else r.clone(), fake.clone(), which)
            b = O.real_fake_loss(r, fake, which)
            vals.append(float(a)); d = max(d, abs(float(a) - float(b)))
    for norm in ("l1", "l2"):
        for alpha, gamma in ((0.0, 0.0), (0.3, 1.0)):
            a = recon_loss(p[:, None], t[:, None], e[:, None], alpha=alpha, gamma=gamma, norm=norm)
            b = O.recon_loss(p, t, e, alpha, gamma, norm)
            vals.append(float(a)); d = max(d, abs(float(a) - float(b)))
    a = fake_generator_loss(fake[:, None]); b = O.fake_generator_loss(fake)
    vals.append(float(a)); d = max(d, abs(float(a) - float(b)))
    W = [real, fake.reshape(4, 4)]
    a = loss_reg_l1(1e-5)(W); b = O.loss_reg_l1(1e-5, W)
    vals.append(float(a)); d = max(d, abs(float(a) - float(b)))
    out["G5_vals"] = np.array(vals, dtype=np.float64)
    pin["G5/losses"] = {"max": d}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims()
    out, pin = {}, {}
    for fn in (gen_G5, gen_G3, gen_G1, gen_G2, gen_G4):
        print("[golden]", fn.__name__, flush=True)
        fn(out, pin)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    worst = max(v for d in pin.values() for v in d.values())
    meta = {"reference": "liupei101/AdvMIL @ v1 (/root/reference)", "torch": torch.__version__,
            "data_seed": DATA_SEED, "param_seed": PARAM_SEED,
            "oracle_vs_reference_maxabs": pin, "worst": worst}
    with open(os.path.join(HERE, "ORACLE_PIN.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(pin, indent=1, sort_keys=True))
    print("worst oracle-vs-reference abs diff:", worst)
    assert worst < 2e-5, worst


if __name__ == "__main__":
    main()
