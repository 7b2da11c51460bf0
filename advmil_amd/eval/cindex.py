"""eval/cindex.py of the reference (concordance_index 10-41, concordance_index_censored 145-198) with the same names, arguments,
return values and exceptions; the O(n^2) pair loop (79-143) runs in the HIP library (advmil_cindex_counts), so the validation /
test passes of every epoch (model_handler.py:278-285) no longer spend seconds in python loops."""
import ctypes

import torch

from .. import _lib


class NoComparablePairException(ValueError):
    """Data of censored event times does not contain one or more comparable pairs (eval/cindex.py:47-50)."""


def _dev(t, device):
    t = torch.as_tensor(t)
    if t.dtype == torch.bool:
        t = t.to(torch.float32)
    return t.to(device=device, dtype=torch.float32).contiguous().reshape(-1)


def concordance_index_censored(event_indicator, event_time, estimate, tied_tol=1e-8, device=None):
    """(cindex, concordant, discordant, tied_risk, tied_time); inputs array-likes or tensors of length n."""
    if device is None:
        device = estimate.device if (torch.is_tensor(estimate) and estimate.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    if not torch.cuda.is_available():
        raise RuntimeError("advmil_amd.eval needs an MI355X: no ROCm device visible (no CPU fallback)")
    ev, tm, est = _dev(event_indicator, device), _dev(event_time, device), _dev(estimate, device)
    n = tm.numel()
    if not (ev.numel() == n == est.numel()):
        raise ValueError("Found input variables with inconsistent numbers of samples")
    if n < 2:
        raise ValueError("Need a minimum of two samples")                       # eval/cindex.py:71-72
    out = torch.empty(6, dtype=torch.int64, device=device)
    _lib.check(_lib.lib().advmil_cindex_counts(ctypes.c_void_p(tm.data_ptr()), ctypes.c_void_p(ev.data_ptr()),
                                               ctypes.c_void_p(est.data_ptr()), n, float(tied_tol), ctypes.c_void_p(out.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "cindex_counts")
    any_event = bool((ev != 0).any().item())
    con, dis, tie, tt, comp, entries = (int(v) for v in out.cpu().tolist())
    if not any_event:
        raise ValueError("All samples are censored")                            # eval/cindex.py:74-75
    if entries == 0:
        raise NoComparablePairException("Data has no comparable pairs, cannot estimate concordance index.")
    cindex = (con + 0.5 * tie) / comp if comp > 0 else float("nan")
    return cindex, con, dis, tie, tt


def concordance_index(y_true, y_pred):
    """y_true[:,0] observed time, y_true[:,1] event indicator; y_pred [n,1] (scalar prediction; the reference negates it into a
    risk) or [n,bins] (discrete hazards -> risk = -sum_k prod_{l<=k} (1 - h_l)). eval/cindex.py:10-41."""
    y_true, y_pred = torch.as_tensor(y_true), torch.as_tensor(y_pred)
    dev = y_pred.device if y_pred.is_cuda else torch.device("cuda", torch.cuda.current_device())
    y_true = y_true.to(dev, torch.float32).reshape(y_true.shape[0], -1)
    y_pred = y_pred.to(dev, torch.float32).reshape(y_pred.shape[0], -1)
    t, e = y_true[:, 0], y_true[:, 1]
    if y_pred.shape[1] == 1:
        return concordance_index_censored(e, t, -y_pred[:, 0], tied_tol=1e-08, device=dev)[0]
    risk = torch.cumprod(1.0 - y_pred, dim=1).sum(dim=1)
    return concordance_index_censored(e, t, -risk, tied_tol=1e-08, device=dev)[0]
