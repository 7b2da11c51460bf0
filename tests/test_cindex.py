"""Concordance index: oracle vs the reference's outputs (CPU, golden fixture) and HIP kernel vs oracle (GPU)."""
import json
import os

import numpy as np
import pytest

from oracle import cindex_oracle as CO
from tests.golden.gen_golden_cindex import CASES, case

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cindex_v1.json")))


def _inputs(idx):
    y_true, pred = case(idx, **CASES[idx])
    ev, tm = y_true[:, 1].astype(bool), y_true[:, 0]
    est = -np.squeeze(pred, axis=1) if pred.shape[1] == 1 else -np.sum(np.cumprod(1.0 - pred, axis=1), axis=1)
    return y_true, pred, ev, tm, est


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_oracle_matches_reference_golden(idx):
    y_true, pred, ev, tm, est = _inputs(idx)
    g = GOLD["cases"][idx]
    c, con, dis, tie, tt = CO.cindex_counts(ev, tm, est)
    assert (con, dis, tie, tt) == (g["concordant"], g["discordant"], g["tied_risk"], g["tied_time"])
    assert abs(c - g["cindex"]) < 1e-15
    assert abs(CO.concordance_index(y_true, pred) - g["top_level"]) < 1e-15


def test_oracle_error_behaviour_matches_reference():
    f32 = np.float32
    with pytest.raises(ValueError):
        CO.cindex_counts(np.array([True]), np.array([1.0], f32), np.array([0.5], f32))
    with pytest.raises(ValueError):
        CO.cindex_counts(np.array([False, False]), np.array([1.0, 2.0], f32), np.array([0.1, 0.2], f32))
    with pytest.raises(CO.NoComparablePairException):
        CO.cindex_counts(np.array([False, True]), np.array([1.0, 2.0], f32), np.array([0.1, 0.2], f32))
    assert GOLD["errors"] == {"one_sample": "ValueError", "all_censored": "ValueError", "no_comparable": "NoComparablePairException"}


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_hip_counts_equal_reference_golden(idx):
    import torch
    from advmil_amd.eval import concordance_index, concordance_index_censored
    y_true, pred, ev, tm, est = _inputs(idx)
    g = GOLD["cases"][idx]
    c, con, dis, tie, tt = concordance_index_censored(torch.from_numpy(ev), torch.from_numpy(tm), torch.from_numpy(est))
    assert (con, dis, tie, tt) == (g["concordant"], g["discordant"], g["tied_risk"], g["tied_time"])     # bit-exact integers
    assert abs(c - g["cindex"]) < 1e-15
    assert abs(concordance_index(torch.from_numpy(y_true), torch.from_numpy(pred)) - g["top_level"]) < 1e-12


@pytest.mark.gpu
def test_hip_counts_equal_oracle_on_larger_inputs_and_errors():
    import torch
    from advmil_amd.eval import NoComparablePairException, concordance_index_censored
    rs = np.random.RandomState(3)
    for n, lv in [(3000, 37), (5001, 0)]:
        tm = rs.rand(n).astype(np.float32)
        if lv:
            tm = (np.floor(tm * lv) / lv).astype(np.float32)
        ev = rs.rand(n) < 0.45
        est = (np.floor(rs.rand(n) * 200) / 200).astype(np.float32)
        want = CO.cindex_counts(ev, tm, est)
        got = concordance_index_censored(torch.from_numpy(ev), torch.from_numpy(tm), torch.from_numpy(est))
        assert got[1:] == want[1:] and abs(got[0] - want[0]) < 1e-15
    with pytest.raises(ValueError):
        concordance_index_censored(torch.tensor([True]), torch.tensor([1.0]), torch.tensor([0.5]))
    with pytest.raises(ValueError):
        concordance_index_censored(torch.tensor([False, False]), torch.tensor([1.0, 2.0]), torch.tensor([0.1, 0.2]))
    with pytest.raises(NoComparablePairException):
        concordance_index_censored(torch.tensor([False, True]), torch.tensor([1.0, 2.0]), torch.tensor([0.1, 0.2]))


@pytest.mark.gpu
def test_hip_full_size_properties():
    """n = 60 000 (3.6e9 pair tests): a perfectly ordered risk gives 1, the reversed one 0, and the counts do not depend on the
    order of the samples."""
    import torch
    from advmil_amd.eval import concordance_index_censored
    n = 60000
    g = torch.Generator().manual_seed(0)
    tm = torch.randperm(n, generator=g).to(torch.float32) / n          # distinct times (float32 rand would repeat a few)
    ev = torch.rand(n, generator=g) < 0.5
    assert concordance_index_censored(ev, tm, -tm)[0] == 1.0
    assert concordance_index_censored(ev, tm, tm)[0] == 0.0
    est = torch.rand(n, generator=g)
    a = concordance_index_censored(ev, tm, est)
    perm = torch.randperm(n, generator=g)
    b = concordance_index_censored(ev[perm], tm[perm], est[perm])
    assert a == b
    assert a[1] + a[2] + a[3] > 0 and abs(a[0] - 0.5) < 0.01
