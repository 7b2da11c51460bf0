#!/usr/bin/env python3
"""Randomised parity of the adversarial step against the oracle (tests/test_handler_variants_gpu.py::run_case: two optimizer steps,
logged losses, predictions, D scores and updated weights at the contract's tolerances): random backbone, bag count, ragged bag
lengths (multiples of 16; some step batches >= 4096 rows so that the padded slab path runs), event flags, label visibility, D loss.
usage: oracle_fuzz.py [cases] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_handler_variants_gpu import run_case  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
for case in range(ncase):
    kind = rnd.choice(("abmil", "abmil", "patch", "cluster"))
    nb = rnd.randint(1, 6)
    big = rnd.random() < 0.5
    lens = tuple(16 * rnd.randint(1, 160 if big else 40) for _ in range(nb))
    if big and sum(lens) < 4096:
        lens = lens[:-1] + (lens[-1] + 16 * ((4096 - sum(lens)) // 16 + rnd.randint(1, 9)),)
    events = tuple(rnd.randint(0, 1) for _ in range(nb))
    mode = rnd.choice(("wlabel", "wlabel", "wolabel"))
    visible = tuple(rnd.random() < 0.6 for _ in range(nb)) if mode == "wolabel" else None
    loss = rnd.choice(("bce", "bce", "hinge", "wasserstein"))
    # hinge / wasserstein: post-Adam quantities only agree to a few 1e-4 (tests/test_handler_variants_gpu.py::test_other_d_losses: Adam
    # moves parameters whose gradient is round-off by +-lr; seed 11's case 26 -- DeepAttMISL, wasserstein -- reached 4.7e-4, and there
    # the fp32 oracle itself is 3.8e-3 off its float64 self on one gradient, test_d_loss_gradients_before_adam). Their raw gradients
    # are held to 2e-5 against the float64 oracle by that test.
    kw = {} if loss == "bce" else {"loss_netD": loss, "tol": 1e-3, "check_weights": False}
    over = {}
    if rnd.random() < 0.35:                          # the other discriminators (concat head, bag-level inner product, projection on y / none)
        over = rnd.choice((dict(disc_type="cat", disc_prj_path=None), dict(disc_prj_iprd="bag"), dict(disc_prj_path="y"),
                           dict(disc_prj_iprd="bag", disc_prj_path=None)))
    kw.update(over)
    t0 = time.time()
    run_case(kind=kind, lens=lens, events=events, visible=visible, mode=mode, **kw)
    print(f"case {case}: {kind} bags {nb} rows/step {sum(lens)} (mod 256 {sum(lens) % 256}) events {events} mode {mode} "
          f"visible {visible} loss {loss} disc {over or 'prj/instance/x'}: ok ({time.time() - t0:.1f} s)", flush=True)
print("all ok")
