"""GPU: the PRODUCT handler under bag-parallel (two ranks sharing the one GPU over gloo) against the single-process run over the same
global step batches, with the SHIPPED DROPOUT RATES ON. World-size invariance (reference semantics
model_handler.py:333-339, 412, 472-478): same dropout masks / generator noise per bag (ops.DeviceRng.rows, parallel.rng_row_maps),
global denominators, summed gradients, all-reduced logs, all-gathered epoch collector."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["abmil", "patch", "cluster", "graph", "abmil-collide", "patch-collide"])
def test_two_rank_step_equals_single_rank_with_dropout_on(kind, tmp_path):
    from tests import dp_worker
    want = dp_worker.run(kind, 1, 0)
    out = str(tmp_path / "r0.pt")
    port = str(29700 + os.getpid() % 1500)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dp_worker", str(r), "2", port, out, kind], cwd=ROOT, env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=500) == 0
    got = torch.load(out, weights_only=False)
    # epoch collector in global bag order
    for k in ("y", "y_hat", "f_fake"):
        a, b = got["cl"][k].double().reshape(-1), want["cl"][k].double().reshape(-1)
        assert a.shape == b.shape and float((a - b).abs().max()) < 2e-6, (k, float((a - b).abs().max()))
    # logged losses: the reduced values, equal to the single-process step's
    assert len(got["logs"]) == len(want["logs"]) == 4
    for la, lb in zip(got["logs"], want["logs"]):
        for key in lb:
            if key == "i_batch":             # the rank's own loader position (local bags seen so far)
                continue
            assert abs(float(la[key]) - float(lb[key])) < 2e-6, (key, la[key], lb[key])
    # weights after two optimizer steps: same update up to the summation order of the all-reduce (Adam amplifies ulp noise where
    # g ~ 0, so compare per-tensor update norms relatively -- a wrong mask would move these by O(1))
    from advmil_amd import synth
    from tests import helpers as H
    kind = kind.split("-")[0]
    for tag, prefix in (("G", f"G-{kind}:"), ("D", "D-prj:")):
        for k in want[tag]:
            p0 = H.T(synth.param(H.PARAM_SEED, prefix + k, tuple(want[tag][k].shape))).double()
            da, db = float((got[tag][k].double() - p0).norm()), float((want[tag][k].double() - p0).norm())
            assert abs(da - db) <= 5e-3 * db + 5e-5, (tag, k, da, db)
            assert float((got[tag][k].double() - want[tag][k].double()).abs().max()) <= 2.5 * 8e-5, (tag, k)
