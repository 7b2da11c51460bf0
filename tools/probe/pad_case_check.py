#!/usr/bin/env python3
"""One case of tools/probe/pad_fuzz.py under the microscope (usage: pad_case_check.py <cases per backbone> <seed> <backbone> <case index>):
the step with the slab pad against the step without it, after ONE optimizer step and after two -- how far the raw gradients of the first
step are apart (the arena keeps them in the eager handler), how far the weights are after Adam, and how many weight entries moved by more
than one Adam sign flip (lr per step). Found by seed 105: DeepAttMISL, 5 bags per step, 10592 / 12048 rows."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import synth  # noqa: E402
from advmil_amd.config import default_cfg  # noqa: E402
from advmil_amd.model import MyHandler  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_parity_gpu import DEV, load_synth  # noqa: E402

ncase, seed, want_kind, want_case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
rnd = random.Random(seed)
for kind in ("abmil", "patch", "cluster"):
    for case in range(ncase):
        bp = rnd.choice((1, 2, 3, 5, 8))
        lens = [16 * rnd.randint(1, 256) for _ in range(2 * bp)]
        while sum(lens[:bp]) < 4096 or sum(lens[bp:]) < 4096:
            lens[rnd.randrange(2 * bp)] += 16 * rnd.randint(32, 200)
        if (kind, case) != (want_kind, want_case):
            continue
        print(kind, "bp", bp, "lens", lens, flush=True)

        def run(pad, nsteps):
            os.environ["ADVMIL_SLAB_PAD"] = str(pad)
            h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=bp, bag_cache_gb=0), device=DEV)
            load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
            h.rng.reset(77)
            loader = []
            for i, n in enumerate(lens[:bp * nsteps]):
                ext = H.T(synth.cluster_ids(0, 600 + i, n)) if kind == "cluster" else torch.zeros(1, 1)
                loader.append((torch.tensor([[i]], dtype=torch.int), [H.bag(600 + i, max(lens))[:, :n].contiguous(), ext], H.label(i)))
            w0 = (h.optimizerG.flat_param.clone(), h.optimizerD.flat_param.clone())
            cl = h._train_each_epoch(loader, "train")
            torch.cuda.synchronize()
            return dict(pred=cl["y_hat"].clone(), gG=h.optimizerG.flat_grad.clone(), gD=h.optimizerD.flat_grad.clone(),
                        wG=h.optimizerG.flat_param.clone(), wD=h.optimizerD.flat_param.clone(), w0=w0, h=h)

        lr = 8e-5
        for nsteps in (1, 2):
            a, b = run(256, nsteps), run(0, nsteps)
            print(f"-- after {nsteps} optimizer step(s): pred max |diff| {float((a['pred'] - b['pred']).abs().max()):.2e}")
            for net in ("G", "D"):
                ga, gb = a["g" + net].double(), b["g" + net].double()
                wa, wb = a["w" + net].double(), b["w" + net].double()
                dg = (ga - gb).abs()
                print(f"   {net}: last raw gradient: max |diff| {float(dg.max()):.2e} of scale {float(gb.abs().max()):.2e} (rel {float(dg.max() / gb.abs().max()):.1e}); "
                      f"entries with |g| < 1e-7: {float((gb.abs() < 1e-7).double().mean()):.3f}; "
                      f"weights: max |diff| {float((wa - wb).abs().max()):.2e} = {float((wa - wb).abs().max()) / lr:.2f} lr, "
                      f"entries > 1e-6: {float(((wa - wb).abs() > 1e-6).double().mean()):.4f}, > 2.05 lr: {int(((wa - wb).abs() > 2.05 * lr).sum())}, "
                      f"> {2.05 * nsteps:.1f} lr: {int(((wa - wb).abs() > 2.05 * nsteps * lr).sum())}")
            if nsteps == 1:
                # which parameters hold the deviating entries
                hh = a["h"]
                # float64 pre-activations of the generator's first layer over the first step's rows: entries within fp32 round-off of the
                # ReLU boundary take either branch depending on the summation order (the pad changes the tiles); the forward does not move
                # (relu(+-1e-8) = 0 to fp32), the backward's mask does, and with it that unit's whole weight-gradient row
                first = [(n, p) for n, p in hh.netG.named_parameters() if p.dim() >= 2 and p.shape[1] == 1024][0]      # the layer applied to the slab
                W0 = first[1].detach()
                w0g = a["w0"][0]
                o0 = int((W0.data_ptr() - hh.optimizerG.flat_param.data_ptr()) // 4)
                Wi = w0g[o0:o0 + W0.numel()].view(W0.shape[0], -1).double()
                bname = first[0].replace("weight", "bias")
                bpar = dict(hh.netG.named_parameters())[bname]
                ob = int((bpar.data_ptr() - hh.optimizerG.flat_param.data_ptr()) // 4)
                bi = w0g[ob:ob + bpar.numel()].double()
                rows = torch.cat([H.bag(600 + i, max(lens))[0, :n] for i, n in enumerate(lens[:bp])], dim=0).to(DEV).double()
                z = rows @ Wi.t() + bi
                near = (z.abs() < 2e-6).nonzero()
                print(f"      float64 pre-activations of {first[0]} over the step's {rows.shape[0]} rows: {near.shape[0]} entries within 2e-6 of 0:",
                      [(int(r), int(c), float(z[r, c])) for r, c in near[:8].tolist()])
                dW = (a["gG"][o0:o0 + W0.numel()] - b["gG"][o0:o0 + W0.numel()]).view(W0.shape[0], -1).abs().max(dim=1).values
                top = torch.topk(dW, 4)
                print("      output units with the largest difference of their weight-gradient row between the two runs:",
                      [(int(i), f"{float(v):.2e}") for v, i in zip(top.values, top.indices)], f"(median over units {float(dW.median()):.1e})")
                d = (a["wG"] - b["wG"]).abs()
                off = 0
                for name, p in hh.netG.named_parameters():
                    k = p.numel()
                    o = int((p.data_ptr() - hh.optimizerG.flat_param.data_ptr()) // 4)
                    frac = float((d[o:o + k] > 1e-6).double().mean())
                    if frac > 0.001:
                        gsl = b["gG"][o:o + k]
                        print(f"      {name} {tuple(p.shape)}: {frac:.4f} of its entries moved; |grad| median {float(gsl.abs().median()):.2e}, max {float(gsl.abs().max()):.2e}")

        # the same bag geometry (lengths, bag and cluster-id seeds, labels) through the ORACLE comparison of the parity suite
        # (tests/test_handler_variants_gpu.py::run_case: two optimizer steps, dropout off, injected noise; losses, predictions, scores and
        # updated weights at the contract's tolerances), with the slab pad and without it
        from tests.test_handler_variants_gpu import run_case
        for pad in (256, 0):
            for part, ll in (("first step's bags", lens[:bp]), ("second step's bags", lens[bp:])):
                os.environ["ADVMIL_SLAB_PAD"] = str(pad)
                try:
                    run_case(kind=kind, lens=tuple(ll), bag_seed0=600)
                    print(f"   oracle comparison, slab pad {pad}, {part} {ll}: ok", flush=True)
                except AssertionError as exc:
                    print(f"   oracle comparison, slab pad {pad}, {part} {ll}: FAIL {str(exc.args[0])[:300]}", flush=True)
