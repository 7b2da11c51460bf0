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
from tests.test_handler_variants_gpu import gradients_before_adam, run_case  # noqa: E402



def relu_boundary_entries(kind, lens, over, band=2e-6):
    """The slab-sized pre-ReLU tensors of the case (G's first layer / region-embedding LayerNorm, D's region-embedding LayerNorm) from the
    float64 parameters and bags of run_case: [(tensor name, row, unit, float64 value)] of the entries within `band` of 0. A 1024-term fp32
    dot product of O(1) values is ~1e-6 away from its float64 value, so such an entry can take the other ReLU branch on ANY fp32 path
    (tools/probe/relu_flip_check.py shows one doing so); the unit's weight gradients then differ by a whole row's contribution."""
    import torch
    import torch.nn.functional as F
    from advmil_amd import synth
    from tests import helpers as H
    PG = {k: H.T(synth.param(H.PARAM_SEED, f"G-{kind}:" + k, tuple(s))).double() for k, s in H.shapes_generator(kind).items()}
    dt = over.get("disc_type", "prj")
    PD = {k: H.T(synth.param(H.PARAM_SEED, ("D-prj:" if dt == "prj" else "D-cat:") + k, tuple(s))).double()
          for k, s in H.shapes_disc(dt, over.get("disc_prj_path", "x")).items()}
    out = []
    for i, n in enumerate(lens):
        x = H.bag(40 + i, max(512, max(lens)))[0, :n].double()
        tensors = []
        if kind in ("abmil", "cluster"):
            pre = "backbone.attention_net.0" if kind == "abmil" else "backbone.phis.0"
            tensors.append((pre, x @ PG[pre + ".weight"].reshape(-1, 1024).t() + PG[pre + ".bias"]))
        else:
            W, b = PG["backbone.patch_embedding_layer.conv.weight"].reshape(-1, 1024), PG["backbone.patch_embedding_layer.conv.bias"]
            tensors.append(("backbone.patch_embedding_layer", F.layer_norm(x @ W.t() + b, (W.shape[0],), PG["backbone.patch_embedding_layer.norm.weight"],
                                                                               PG["backbone.patch_embedding_layer.norm.bias"], 1e-5)))
        W, b = PD["net_pair_one.embedding.conv.weight"].reshape(-1, 1024), PD["net_pair_one.embedding.conv.bias"]
        tensors.append(("net_pair_one.embedding", F.layer_norm(x @ W.t() + b, (W.shape[0],), PD["net_pair_one.embedding.norm.weight"],
                                                                PD["net_pair_one.embedding.norm.bias"], 1e-5)))
        for name, t in tensors:
            for r, u in (t.abs() < band).nonzero().tolist():
                out.append((f"bag {i}: {name}", r, u, float(t[r, u])))
    return out


def log_counted(rec):
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fuzz_counted_cases.jsonl", "a") as fh:
        fh.write(json.dumps(rec) + "\n")


counted = []
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
    try:
        run_case(kind=kind, lens=lens, events=events, visible=visible, mode=mode, **kw)
    except AssertionError as exc:
        # The post-Adam weight comparison allows 0.01 % of a tensor's entries to sit one sign flip (<= 2 lr per step) away from the
        # oracle. A case that exceeds that COUNT while every entry stays inside the sign-flip bound is accepted only if the raw
        # gradients of the same configuration (no Adam in between) agree with the float64 oracle at 2e-5 of each tensor's scale;
        # it is then counted, logged with what reproduces it, and the number of such cases is bounded.
        a = exc.args[0] if exc.args else None
        is_count = isinstance(a, tuple) and len(a) >= 4 and isinstance(a[1], int) and a[1] <= a[2] // 1000 and a[3] < 2.05 * 8e-5 * 2
        if not is_count:
            raise
        base = {"tool": "tools/probe/oracle_fuzz.py", "argv": sys.argv[1:], "case": case, "kind": kind, "lens": lens, "events": events,
                "mode": mode, "visible": visible, "loss": loss, "disc": over, "bag_seeds": [40 + i for i in range(len(lens))],
                "tensor": a[0], "entries_off": a[1], "of": a[2], "max_abs": a[3]}
        try:
            gradients_before_adam(loss, kind, lens, events, visible, **over)
            why = ("post-Adam sign flips beyond 0.01 % of the entries, all within 2 lr per step; raw gradients of the same configuration "
                   "verified at 2e-5 (tests/test_handler_variants_gpu.py::gradients_before_adam)")
        except AssertionError as gexc:
            near = relu_boundary_entries(kind, lens, over)
            if not near:
                raise
            why = (f"ReLU-boundary flip: raw gradient check failed with {gexc.args[0] if gexc.args else gexc}, and the float64 pre-activations hold "
                   f"{len(near)} entries within 2e-6 of 0: {near[:4]} (a property of the input: tools/probe/relu_flip_check.py)")
            base["relu_boundary_entries"] = near[:8]
        counted.append(case)
        assert len(counted) <= max(1, ncase // 15), "too many counted cases"
        base["why_counted"] = why
        log_counted(base)
        print(f"case {case}: {kind} bags {nb} rows/step {sum(lens)} loss {loss} disc {over or 'prj/instance/x'}: {a[1]} of {a[2]} entries of {a[0]} "
              f"beyond the sign-flip count (max {a[3]:.1e}): counted -- {why[:110]}...", flush=True)
        continue
    print(f"case {case}: {kind} bags {nb} rows/step {sum(lens)} (mod 256 {sum(lens) % 256}) events {events} mode {mode} "
          f"visible {visible} loss {loss} disc {over or 'prj/instance/x'}: ok ({time.time() - t0:.1f} s)", flush=True)
print("all ok")
