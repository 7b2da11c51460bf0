"""The five helpers of the reference's utils/func.py that sit on the G+D step path
(SURVEY.md §2 #12): generate_noise (154-164), collect_tensor (112-125), agg_tensor (127-133),
sparse_key / sparse_str (135-152), seed_everything (166-175). Same names and argument meaning."""
import random

import numpy as np
import torch

from .. import ops


def generate_noise(*dims, to_device="cuda", distribution="uniform", rng=None):
    """[dims] noise drawn ON the device by the counter RNG (the reference samples on the CPU and copies:
    utils/func.py:154-164). uniform -> U[0,1); gaussian -> N(0,1) by Box-Muller on two uniform draws."""
    assert distribution in ["uniform", "gaussian"]
    rng = rng or ops.default_rng(to_device)
    n = int(np.prod(dims))
    w = int(dims[-1])
    if distribution == "uniform":
        return rng.uniform(n, "noise", width=w).reshape(*dims)
    u1 = rng.uniform(n, "noise_u1", width=w).clamp_min(2.0 ** -24)
    u2 = rng.uniform(n, "noise_u2", width=w)
    return (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)).reshape(*dims)


def collect_tensor(collector, real, fake):
    for key, val in (("real", real), ("fake", fake)):
        if val is not None:
            collector[key] = val if collector[key] is None else torch.cat([collector[key], val], dim=0)
    return collector


def agg_tensor(collector, data):
    for k, v in data.items():
        prev = collector.get(k)
        collector[k] = v if prev is None else torch.cat([prev, v], dim=0)
    return collector


def sparse_key(d, prefixes: str = ""):
    """cfg keys starting with `prefixes` + '_' -> dict without the prefix (func.py:135-146)."""
    if prefixes == "":
        return d
    out = {}
    for k, v in d.items():
        if k.startswith(prefixes):
            rest = k.split(prefixes)[1]
            if len(rest) >= 2:
                out[rest[1:]] = v
    return out


def sparse_str(s, sep="-", dtype=int):
    return [s] if not isinstance(s, str) else [dtype(tok) for tok in s.split(sep)]


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
        ops.default_rng(torch.device("cuda", torch.cuda.current_device())).reset(seed)


def dropout_small(x, p, training, rng, tag=""):
    """Dropout for the [1,d]-sized head tensors, drawn from the same counter RNG as the kernels
    (so a parity test can regenerate the mask: advmil_amd.synth.dropout_keep)."""
    if not training or p <= 0.0:
        return x
    return ops.dropout(x, p, rng, tag)
