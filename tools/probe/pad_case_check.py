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
                d = (a["wG"] - b["wG"]).abs()
                off = 0
                for name, p in hh.netG.named_parameters():
                    k = p.numel()
                    o = int((p.data_ptr() - hh.optimizerG.flat_param.data_ptr()) // 4)
                    frac = float((d[o:o + k] > 1e-6).double().mean())
                    if frac > 0.001:
                        gsl = b["gG"][o:o + k]
                        print(f"      {name} {tuple(p.shape)}: {frac:.4f} of its entries moved; |grad| median {float(gsl.abs().median()):.2e}, max {float(gsl.abs().max()):.2e}")
