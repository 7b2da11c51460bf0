#!/usr/bin/env python3
"""Golden vectors for the concordance index: runs the REFERENCE's eval/cindex.py on seeded cases and pins
oracle/cindex_oracle.py against it count for count. Build container only. Usage: python tests/golden/gen_golden_cindex.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from advmil_amd import synth  # noqa: E402
from oracle import cindex_oracle as CO  # noqa: E402


def case(idx, n, time_levels, est_levels, p_event, bins=0):
    """Seeded inputs (counter RNG): times / estimates quantised to a few levels so that ties in both occur."""
    u = synth.device_uniform(900 + idx, 1, 3 * n + n * max(bins, 1)).astype(np.float32)
    t = u[:n]
    t = np.floor(t * time_levels).astype(np.float32) / np.float32(time_levels) if time_levels else t
    e = (u[n:2 * n] < p_event).astype(np.float32)
    if bins:
        pred = (0.05 + 0.9 * u[3 * n:3 * n + n * bins]).reshape(n, bins).astype(np.float32)
    else:
        p = u[2 * n:3 * n]
        p = np.floor(p * est_levels).astype(np.float32) / np.float32(est_levels) if est_levels else p
        pred = p.reshape(n, 1).astype(np.float32)
    return np.stack([t, e], axis=1).astype(np.float32), pred


CASES = [dict(n=2, time_levels=0, est_levels=0, p_event=1.0), dict(n=17, time_levels=5, est_levels=4, p_event=0.6),
         dict(n=64, time_levels=0, est_levels=0, p_event=0.5), dict(n=200, time_levels=12, est_levels=9, p_event=0.4),
         dict(n=333, time_levels=3, est_levels=0, p_event=0.8), dict(n=1000, time_levels=50, est_levels=40, p_event=0.3),
         dict(n=150, time_levels=10, est_levels=0, p_event=0.5, bins=4), dict(n=40, time_levels=1, est_levels=3, p_event=0.5)]


def main():
    from eval import cindex as R          # the reference module (needs sklearn, present here)
    out = []
    for idx, c in enumerate(CASES):
        y_true, pred = case(idx, **c)
        ev, tm = y_true[:, 1].astype(bool), y_true[:, 0]
        if pred.shape[1] == 1:
            est = -np.squeeze(pred, axis=1)
        else:
            est = -np.sum(np.cumprod(1.0 - pred, axis=1), axis=1)
        want = R.concordance_index_censored(ev, tm, est, tied_tol=1e-08)
        got = CO.cindex_counts(ev, tm, est)
        assert tuple(int(v) for v in want[1:]) == tuple(int(v) for v in got[1:]), (idx, want, got)
        assert abs(want[0] - got[0]) < 1e-15, (idx, want[0], got[0])
        top = R.concordance_index(y_true.copy(), pred.copy())
        assert abs(top - CO.concordance_index(y_true, pred)) < 1e-15
        out.append({"case": c, "cindex": float(want[0]), "concordant": int(want[1]), "discordant": int(want[2]),
                    "tied_risk": int(want[3]), "tied_time": int(want[4]), "top_level": float(top)})
        print(idx, c, out[-1]["cindex"], want[1:])
    # error behaviour of the reference, recorded as facts
    errs = {}
    for name, (ev, tm, est) in {"one_sample": ([True], [1.0], [0.5]), "all_censored": ([False, False], [1.0, 2.0], [0.1, 0.2]),
                                "no_comparable": ([False, True], [1.0, 2.0], [0.1, 0.2])}.items():
        try:
            R.concordance_index_censored(np.array(ev), np.array(tm, dtype=np.float32), np.array(est, dtype=np.float32))
            errs[name] = None
        except Exception as exc:
            errs[name] = type(exc).__name__
        try:
            CO.cindex_counts(np.array(ev), np.array(tm, dtype=np.float32), np.array(est, dtype=np.float32))
            mine = None
        except Exception as exc:
            mine = type(exc).__name__
        assert mine == errs[name], (name, mine, errs[name])
    json.dump({"cases": out, "errors": errs, "reference": "liupei101/AdvMIL@v1 eval/cindex.py"}, open(os.path.join(HERE, "cindex_v1.json"), "w"), indent=1)
    print("errors", errs)


if __name__ == "__main__":
    main()
