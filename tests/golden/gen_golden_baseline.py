#!/usr/bin/env python3
"""Golden vectors G7 for the supervised baselines: the REFERENCE's SurvNet (model/BaseSurv.py), its losses
(loss/utils.py MSE_loss / SurvMLE / SurvPLE / recon_loss) and two optimizer steps through its own
BaselineHandler._update_network (model/baseline_handler.py:328-368), and the pin of oracle/advmil_oracle.py::baseline_step against
them. Build container only (reuses the shims of gen_golden.py). Usage: python tests/golden/gen_golden_baseline.py"""
import json
import os
import sys
from functools import partial
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402  (shims, seeded inputs, helpers)

from advmil_amd import synth  # noqa: E402
from oracle import advmil_oracle as O  # noqa: E402

# (tag, backbone, task, out_dim, out_scale)
CONFIGS = [("abmil_reg", "abmil", "surv_reg", 1, "sigmoid"), ("cluster_cox", "cluster", "surv_cox", 1, "none"),
           ("abmil_nll", "abmil", "surv_nll", 4, "sigmoid"), ("patch_reg", "patch", "surv_reg", 1, "sigmoid")]
NB, N, STEPS = 8, 512, 2
LR, WD, L1 = 8e-5, 5e-4, 1e-5


def labels(task, i):
    y = GG.T(synth.label(GG.DATA_SEED, i)).clone()            # [1,2]: t in (0,1), e
    if task == "surv_nll":
        y[0, 0] = float(int(y[0, 0] * 4) % 4)                 # bin index 0..3 (time_bins: 4, cfg_nlst.yaml:18)
    elif task == "surv_cox":
        y[0, 0] = y[0, 0] * 100.0 + i * 0.01                  # 'origin' time format, no ties (SurvPLE assumes none)
    return y


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    GG.install_shims()
    from loss.utils import MSE_loss, SurvMLE, SurvPLE, loss_reg_l1, recon_loss
    from model.backbone import load_backbone
    from model.BaseSurv import SurvNet
    from model.baseline_handler import BaselineHandler
    from optim import create_optimizer
    out, pin = {}, {}
    for tag, kind, task, dout, oscale in CONFIGS:
        net = SurvNet(384, dout, load_backbone(kind, [1024, 384, 384]), hops=1, norm=False, dropout=0.6, out_scale=oscale)
        P = GG.load_synth(net, prefix=f"S-{tag}:")
        GG.zero_dropout(net)
        h = object.__new__(BaselineHandler)                   # the step method only needs these attributes (328-368)
        h.net, h.bcb = net, kind
        if task == "surv_nll":
            h.supervised_loss = SurvMLE(alpha=0.0)
        elif task == "surv_cox":
            h.supervised_loss = SurvPLE()
        elif kind == "patch":
            h.supervised_loss = partial(MSE_loss, include_censored=False)
        else:
            h.supervised_loss = partial(recon_loss, alpha=0.0, gamma=0.0, norm="l1")
        h.loss_l1 = loss_reg_l1(L1)
        h.optimizer = create_optimizer(SimpleNamespace(opt="adam", weight_decay=WD, lr=LR, opt_eps=None, opt_betas=None, momentum=None), net)
        st, oP = {}, P
        logs_all, preds_all, worst = [], [], 0.0
        for s in range(STEPS):
            xs, ys, bags = [], [], []
            for j in range(NB):
                i = s * NB + j
                x = GG.T(synth.bag(GG.DATA_SEED, 100 + i, N))
                ext = GG.T(synth.cluster_ids(GG.DATA_SEED, 100 + i, N)) if kind == "cluster" else torch.zeros(1, 1)
                y = labels(task, i)
                xs.append([x, ext]); ys.append(y); bags.append((x, ext if kind == "cluster" else None, y))
            GG.LOG.clear()
            cur = BaselineHandler._update_network(h, s + 1, xs, ys)
            lg = dict(GG.LOG[-1])
            oP, olog, opreds = O.baseline_step(oP, st, bags, kind, task, oscale, 1, None, LR, WD, L1, (0.0, 0.0, "l1"), 0.0, False)
            worst = max(worst, abs(lg["train_batch/net/loss_supervision"] - olog["loss_supervision"]),
                        abs(lg["train_batch/net/loss_total"] - olog["loss_total"]), GG.maxdiff(cur.detach(), torch.cat(opreds)))
            logs_all.append([lg["train_batch/net/loss_supervision"], lg["train_batch/net/loss_total"]])
            preds_all.append(cur.detach().numpy().copy())
        ref = {k: v.detach().clone() for k, v in net.state_dict().items()}
        worst_w = max(GG.maxdiff(ref[k], oP[k]) for k in ref)
        pin[f"G7/{tag}"] = {"logs_preds": worst, "weights_after_2_steps": worst_w}
        out[f"G7_{tag}_logs"] = np.array(logs_all, dtype=np.float64)
        out[f"G7_{tag}_preds"] = np.stack(preds_all)
        keys = sorted(ref)
        out[f"G7_{tag}_keys"] = np.array(keys)
        out[f"G7_{tag}_dstats"] = np.array([[float((ref[k].double() - P[k].double()).sum()), float((ref[k].double() - P[k].double()).norm())] for k in keys])
        print(tag, pin[f"G7/{tag}"], logs_all, flush=True)
    np.savez_compressed(os.path.join(HERE, "golden_baseline_v1.npz"), **out)
    json.dump({"reference": "liupei101/AdvMIL @ v1", "oracle_vs_reference_maxabs": pin,
               "worst": max(v for d in pin.values() for v in d.values())}, open(os.path.join(HERE, "ORACLE_PIN_baseline.json"), "w"), indent=1)
    assert max(v for d in pin.values() for v in d.values()) < 5e-5


if __name__ == "__main__":
    main()
