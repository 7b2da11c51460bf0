#!/usr/bin/env python3
"""Randomised parity of the supervised baselines (BaselineHandler) against the oracle's baseline_step (pinned to the
reference's BaselineHandler._update_network by golden G7): random backbone (ABMIL / DeepAttMISL with ANY bag length, ESAT with
multiples of 16), task (regression / Cox partial likelihood / discrete-time NLL), bag count, ragged lengths (some step batches >= 4096
rows: padded slab), two optimizer steps; predictions and logged losses at 2e-5, weights like tests/test_handler_variants_gpu.py.
usage: baseline_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import synth  # noqa: E402
from advmil_amd.config import default_baseline_cfg  # noqa: E402
from advmil_amd.model import BaselineHandler  # noqa: E402
from oracle import advmil_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_parity_gpu import DEV, load_synth, zero_dropout  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def relu_boundary_entries(P, bags, kind, task, band=2e-6):
    """Re-run the forward + backward of ONE oracle step in float64 (a fresh optimizer state: only the forward matters) with every torch.relu call recorded: [(call index, flat position, value)] of the ReLU inputs within
    `band` of zero. Such an input takes either branch depending on the fp32 summation order (tools/probe/relu_flip_check.py)."""
    found, calls, real = [], [0], torch.relu

    def spy(x):
        if x.dtype == torch.float64 and x.numel() > 1:
            for pos in (x.detach().abs() < band).flatten().nonzero().flatten().tolist()[:4]:
                found.append((calls[0], pos, float(x.detach().flatten()[pos])))
        calls[0] += 1
        return real(x)
    dbl = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t      # noqa: E731
    P64 = {k_: dbl(v_) for k_, v_ in P.items()}
    b64 = [(x.double(), None if e is None else e.double(), y.double()) for x, e, y in bags]
    torch.relu = spy
    try:
        O.baseline_step(P64, {}, b64, kind=kind, task=task, out_scale="none" if task == "surv_cox" else "sigmoid")
    finally:
        torch.relu = real
    return found


for case in range(ncase):
    kind = rnd.choice(("abmil", "cluster", "patch"))
    task = rnd.choice(("surv_reg", "surv_cox", "surv_nll"))
    pdh = "384-4" if task == "surv_nll" else "384-1"
    nb = rnd.randint(1, 5)
    unit = 16 if kind == "patch" else 1
    big = rnd.random() < 0.4
    lens = [unit * rnd.randint(1, (2600 if big else 700) // unit) for _ in range(nb)]
    if big and sum(lens) < 4096:
        lens[-1] += unit * ((4200 - sum(lens)) // unit + 1)
    h = BaselineHandler(default_baseline_cfg(bcb_mode=kind, task=task, pdh_dims=pdh, bp_every_batch=nb, bag_cache_gb=0), device=DEV)
    P = load_synth(h.net, f"S-fz{case}:")
    zero_dropout(h.net)
    bags, loader = [], []
    for s in range(2):
        for j, n in enumerate(lens):
            i = s * nb + j
            x = H.bag(900 + i, max(lens))[:, :n].contiguous()
            y = H.label(i).clone()
            if task == "surv_nll":
                y[0, 0] = float(int(y[0, 0] * 4) % 4)
            elif task == "surv_cox":
                y[0, 0] = y[0, 0] * 100.0 + i * 0.01
            ext = H.T(synth.cluster_ids(0, 900 + i, n)) if kind == "cluster" else None
            bags.append((x, ext, y))
            loader.append((torch.tensor([[i]], dtype=torch.int), [x, ext if ext is not None else torch.zeros(1, 1)], y))
    cl = h._train_each_epoch(loader, "train")
    logs = h.pop_logs()
    st = {}
    suspect = False
    for s in range(2):
        if s == 1:           # kept for the float64 re-run of a suspect second step (relu_boundary_entries)
            P1 = {k_: v_.clone() for k_, v_ in P.items()}
        P, lg, preds = O.baseline_step(P, st, bags[s * nb:(s + 1) * nb], kind=kind, task=task,
                                       out_scale="none" if task == "surv_cox" else "sigmoid")
        want = torch.cat(preds, dim=0)
        got = cl["y_hat"][s * nb:(s + 1) * nb]
        ep = float((got - want).abs().max())
        el = max(abs(logs[s]["train_batch/net/loss_supervision"] - lg["loss_supervision"]), abs(logs[s]["train_batch/net/loss_total"] - lg["loss_total"]))
        sc = max(1.0, abs(lg["loss_total"]))
        if s == 1 and not (ep < 2e-5 * max(1.0, float(want.abs().max())) and el < 2e-5 * sc) and ep < 1e-3 and el < 1e-3:
            # first step exact, second step 1e-4 off: one ReLU pre-activation (FFN / region embedding) within fp32 round-off of zero took
            # the other branch on one side -- its whole gradient contribution moves, Adam turns that into +-lr on many weights
            # (seed 8, case 18: ESAT, 255 tokens: one FFN unit of one token; every raw gradient upstream 5e-4 .. 4.5e-3 off float64
            # while the forward agrees to 2e-7; with other parameters or bags nothing deviates). Counted, not failed.
            suspect = True
            break
        assert ep < 2e-5 * max(1.0, float(want.abs().max())) and el < 2e-5 * sc, (case, kind, task, lens, s, ep, el)
    if suspect:
        # accepted only if the float64 evaluation of that second step really holds a ReLU input within 2e-6 of zero
        near = relu_boundary_entries(P1, bags[nb:2 * nb], kind, task)
        assert near, (case, kind, task, lens, "second step off by", ep, el, "and no ReLU input within 2e-6 of zero in float64")
        nsus = globals().get("nsus", 0) + 1
        globals()["nsus"] = nsus
        assert nsus <= max(1, ncase // 15), "too many second-step deviations to be ReLU-boundary flips"
        print(f"case {case}: {kind} {task} lens {lens}: second step off by {ep:.1e} (ReLU boundary flip, see source): counted", flush=True)
        # every counted case is logged with what reproduces it (profiles/r0x_fuzz_counted_cases.jsonl: VERDICT r3 weak #2)
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/fuzz_counted_cases.jsonl", "a") as fh:
            fh.write(json.dumps({"tool": "tools/probe/baseline_fuzz.py", "argv": sys.argv[1:], "case": case, "kind": kind, "task": task,
                                 "pdh_dims": pdh, "lens": lens, "bag_seeds": [900 + i for i in range(2 * nb)], "param_prefix": f"S-fz{case}:",
                                 "second_step_pred_dev": ep, "second_step_loss_dev": el, "relu_inputs_within_2e-6_of_zero_in_float64": near[:8]}) + "\n")
        continue
    for k, v in h.net.state_dict().items():
        diff = (v.cpu() - P[k]).abs()
        n_off = int((diff >= 5e-5).sum())
        assert n_off <= max(1, diff.numel() // 5000) and float(diff.max()) < 2.05 * 8e-5 * 2, (case, kind, task, k, n_off, float(diff.max()))
    print(f"case {case}: {kind} {task} lens {lens} (rows/step {sum(lens)}): ok", flush=True)
print("all ok")
