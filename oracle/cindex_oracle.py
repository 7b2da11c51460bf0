"""ORACLE — TEST INFRASTRUCTURE ONLY. Not product code.

numpy restatement of the reference's concordance index (eval/cindex.py:10-41 `concordance_index`, 79-143
`_get_comparable` / `_estimate_concordance_index`, the scikit-survival estimator with unit weights). Only `tests/` may import
this module. Pinned: `tests/golden/gen_golden_cindex.py` runs the reference's own function on seeded inputs (ties in time, ties
in the estimate, all-censored tie groups) and checks this restatement against it count for count; the outputs are committed as
`tests/golden/cindex_v1.json`.

Definition restated (eval/cindex.py:79-105): sample i with an event is compared with every j whose time is strictly later, and
with the CENSORED samples sharing its time (`tied_time` counts those). With estimate = -prediction (cindex.py:35,41), a
comparable pair is a tie in risk if |est_j - est_i| <= tied_tol, else concordant if est_j < est_i, else discordant
(cindex.py:125-137); cindex = (concordant + 0.5 * tied_risk) / comparable. Comparisons run in the dtype of the inputs
(float32 from the handlers), as numpy does for the reference.
"""
import numpy as np


class NoComparablePairException(ValueError):
    pass


def cindex_counts(event, time, estimate, tied_tol=1e-8):
    """(cindex, concordant, discordant, tied_risk, tied_time) — eval/cindex.py:107-143."""
    event = np.asarray(event).astype(bool)
    time = np.asarray(time)
    est = np.asarray(estimate)
    n = len(time)
    if n < 2:
        raise ValueError("Need a minimum of two samples")                      # cindex.py:71-72
    if not event.any():
        raise ValueError("All samples are censored")                           # cindex.py:74-75
    tol = est.dtype.type(tied_tol) if np.issubdtype(est.dtype, np.floating) else tied_tol
    con = dis = tie = tt = comp = 0
    for i in np.nonzero(event)[0]:
        later = time > time[i]
        same_cens = (time == time[i]) & ~event
        mask = later | same_cens                                                # cindex.py:96-101
        tt += int(same_cens.sum())
        if not mask.any() and not (later.any() or same_cens.any()):
            pass
        e = est[mask]
        ties = np.absolute(e - est[i]) <= tol                                   # cindex.py:125
        c = (e < est[i]) & ~ties                                                # cindex.py:128-129
        con += int(c.sum()); tie += int(ties.sum()); dis += int(e.size - c.sum() - ties.sum())
        comp += int(mask.sum())
    # the reference registers an event as "comparable" even when its mask is empty (cindex.py:94-101) as long as it is not the
    # last sorted sample; it raises only if NO event got an entry, i.e. every event sits in the last time group with nothing after
    order = np.argsort(time, kind="stable")
    has_entry = False
    i = 0
    while i < n - 1:
        end = i + 1
        while end < n and time[order[end]] == time[order[i]]:
            end += 1
        if event[order[i:end]].any():
            has_entry = True
        i = end
    if not has_entry:
        raise NoComparablePairException("Data has no comparable pairs, cannot estimate concordance index.")
    cindex = (con + 0.5 * tie) / comp if comp > 0 else float("nan")
    return cindex, con, dis, tie, tt


def concordance_index(y_true, y_pred):
    """eval/cindex.py:10-41: y_true[:,0] time, y_true[:,1] event; y_pred [n,1] (risk-like scalar) or [n,bins] (hazards)."""
    y_true = np.asarray(y_true)
    y_pred = np.asarray(y_pred)
    if y_pred.shape[1] == 1:
        yt, yp = np.squeeze(y_true), np.squeeze(y_pred)
        return cindex_counts(yt[:, 1].astype(bool), yt[:, 0], -yp)[0]
    survival = np.cumprod(1.0 - y_pred, axis=1)
    risk = np.sum(survival, axis=1)
    return cindex_counts(y_true[:, 1].astype(bool), y_true[:, 0], -risk)[0]
