"""Counter-based synthetic data: bags, labels, cluster ids, k-NN graphs and parameter tensors.

Everything is a pure function of (seed, tag, element index) through splitmix64, so the
CPU oracle, the golden-fixture generator (which runs beside the reference in the build
container) and the GPU box all see bit-identical inputs without shipping 33 MB bags.
The kernels' own dropout / generator-noise draw (csrc/common.h: a splitmix64 key per call site, a
32-bit mixer per element) is restated here as `kernel_hash32` / `dropout_keep` / `kernel_uniform`,
which lets a parity test regenerate a kernel's dropout mask on the host.

Bag layout follows the reference loader (`dataset/PatchWSI.py:70-83`): one sample is
`(idx[1,1] int32, (x[1,N,1024] f32, ext), y[1,2] = (t, e) f32)`.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """One splitmix64 output for each uint64 counter in `x` (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, tag) -> np.uint64:
    """Key of an independent stream: tag is an int or a string (crc32'd)."""
    if isinstance(tag, str):
        tag = zlib.crc32(tag.encode())
    a = splitmix64(np.uint64(int(tag) & 0xFFFFFFFFFFFFFFFF))
    return splitmix64(np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) ^ a)


def _counters(key, n, offset=0):
    with np.errstate(over="ignore"):
        return np.uint64(key) + np.arange(offset, offset + n, dtype=np.uint64)


def uniform01(key, n, offset=0):
    """float32 U[0,1) with 24 random bits: (hash >> 40) * 2^-24 — the in-kernel formula."""
    z = splitmix64(_counters(key, n, offset))
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def normal(key, n):
    """float32 N(0,1) by Box-Muller on two decorrelated streams."""
    z1 = splitmix64(_counters(key, n))
    z2 = splitmix64(z1 ^ np.uint64(0xD1B54A32D192ED03))
    u1 = ((z1 >> np.uint64(11)).astype(np.float64) + 1.0) * (2.0 ** -53)
    u2 = (z2 >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def kernel_hash32(key, n, offset=0):
    """csrc/common.h::rng_hash32 for the element indices offset .. offset+n-1 of the stream `key` (uint64) -> uint32 array."""
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    klo, khi = np.uint32(key & 0xFFFFFFFF), np.uint32(key >> 32)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) + klo
        x ^= x >> np.uint32(16); x *= np.uint32(0x21F0AAAD)
        x ^= khi ^ ((idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77))
        x ^= x >> np.uint32(15); x *= np.uint32(0x735A2D97)
        x ^= x >> np.uint32(15)
    return x


def kernel_uniform01(key, n, offset=0):
    """float32 U[0,1) of the kernels' per-element draw: (rng_hash32 >> 8) * 2^-24."""
    return (kernel_hash32(key, n, offset) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def dropout_keep(seed, stream, n, p, offset=0):
    """Host restatement of the kernels' dropout decision: keep element i iff u_i >= p.

    `seed` is the step seed held in device memory, `stream` the per-call-site stream id;
    the kernel hashes (mix(seed, stream), i) with the 32-bit element mixer (csrc/common.h: rng_key / rng_hash32 / rng_uniform).
    """
    key = rng_key(seed, stream)
    return kernel_uniform01(key, n, offset) >= np.float32(p)


def rng_key(seed, stream) -> np.uint64:
    """csrc/rng.h::rng_key — splitmix64(seed ^ splitmix64(stream))."""
    a = splitmix64(np.uint64(int(stream) & 0xFFFFFFFFFFFFFFFF))
    return splitmix64(np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) ^ a)


def attn_dropout_keep(seed, stream, row_ids, nhead, head, n_keys, p):
    """Host restatement of the fused attention kernels' dropout decision (csrc/attn.hip): query with global region row id r
    (row_ids, array), head `head`, key j in [0, n_keys) is kept iff byte (j & 3) of
        mix(rowkey + (j >> 2) * 0x9E3779B9)  is  >= floor(p * 256),      mix(x): x ^= x >> 15; x *= 0x7feb352d; x ^= x >> 15,
    rowkey = high 32 bits of splitmix64(key(seed, stream) + r*nhead + head): one 32-bit hash per query and group of 4 keys.
    -> bool [len(row_ids), n_keys]. The kernels scale the kept probabilities by 256 / (256 - floor(256 p))."""
    key = rng_key(seed, stream)
    with np.errstate(over="ignore"):
        rid = np.asarray(row_ids, dtype=np.uint64) * np.uint64(nhead) + np.uint64(head)
        rk = (splitmix64(np.uint64(key) + rid) >> np.uint64(32)).astype(np.uint32)
        j = np.arange(n_keys, dtype=np.uint32)
        x = rk[:, None] + (j >> np.uint32(2))[None, :] * np.uint32(0x9E3779B9)
        x ^= x >> np.uint32(15); x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        byte = (x >> ((j & np.uint32(3)) * np.uint32(8))[None, :]) & np.uint32(0xFF)
    return byte >= np.uint32(int(float(np.float32(p)) * 256.0))


def attn_dropout_scale(p):
    """1 / (1 - p_eff) of the attention kernels: p quantised to 1/256."""
    return 256.0 / (256.0 - int(float(np.float32(p)) * 256.0))


def kernel_uniform(seed, stream, n, offset=0):
    """U[0,1) exactly as the kernels draw it (generator noise, `utils/func.py:154-164`; advmil_uniform_fill)."""
    return kernel_uniform01(rng_key(seed, stream), n, offset)


def device_uniform(seed, stream, n, offset=0):
    """U[0,1) vectors the golden fixtures were generated from (injected generator noise, C-index inputs): splitmix64 per element,
    which is what the kernels drew in round 1. Kept bit for bit -- the fixtures' inputs are regenerated from it; the kernels' own
    draw is `kernel_uniform`."""
    return uniform01(rng_key(seed, stream), n, offset)


# ---------------------------------------------------------------------------------------
# bags / labels
# ---------------------------------------------------------------------------------------
def bag(seed: int, idx: int, n_patches: int, channels: int = 1024) -> np.ndarray:
    """x ~ N(0,1), float32 [1, N, C] (SURVEY §8d synthetic inputs)."""
    key = stream_key(seed, ("bag", idx, n_patches, channels).__repr__())
    return normal(key, n_patches * channels).reshape(1, n_patches, channels)


def label(seed: int, idx: int) -> np.ndarray:
    """y[1,2] = (t ~ U(0,1), e = idx mod 2) — `time_format: ratio`, cfg_nlst.yaml:17."""
    t = uniform01(stream_key(seed, ("label", idx).__repr__()), 1)[0]
    # keep t away from 0/1 so |p-t| and relu(t-p) have non-degenerate gradients
    t = np.float32(0.05 + 0.9 * t)
    return np.array([[t, np.float32(idx % 2)]], dtype=np.float32)


def cluster_ids(seed: int, idx: int, n_patches: int, n_clusters: int = 8) -> np.ndarray:
    """float32 ids in {0..7}, shape [N] (`dataset/PatchWSI.py:93-95` hands floats over)."""
    u = uniform01(stream_key(seed, ("cluster", idx, n_patches).__repr__()), n_patches)
    return np.minimum((u * n_clusters).astype(np.int64), n_clusters - 1).astype(np.float32)


def grid_knn_graph(n_patches: int, k: int = 8) -> np.ndarray:
    """edge_index[2, k*N] int64: patches on a ceil(sqrt N) grid, k nearest (8-neighbourhood,
    clamped at the border by reflecting to the nearest valid cells). Mirrors the edge layout of
    `tools/patchgcn_graph_s2.py:66-80`: source = repeat(range(N), k), target = neighbour."""
    side = int(np.ceil(np.sqrt(n_patches)))
    ii = np.arange(n_patches)
    r, c = ii // side, ii % side
    offs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)][:k]
    tgt = np.empty((n_patches, len(offs)), dtype=np.int64)
    for j, (dr, dc) in enumerate(offs):
        rr = np.clip(r + dr, 0, side - 1)
        cc = np.clip(c + dc, 0, side - 1)
        t = rr * side + cc
        t = np.where(t >= n_patches, ii, t)
        tgt[:, j] = t
    src = np.repeat(ii, len(offs))
    return np.stack([src, tgt.reshape(-1)]).astype(np.int64)


def sample(seed: int, idx: int, n_patches: int, mode: str = "abmil"):
    """One loader sample as numpy arrays: (idx[1,1] i32, (x, ext), y[1,2])."""
    x = bag(seed, idx, n_patches)
    if mode == "cluster":
        ext = cluster_ids(seed, idx, n_patches)[None, :]      # default_collate adds the batch dim
    else:
        ext = np.zeros((1, 1), dtype=np.float32)              # torch.Tensor([0]) collated
    return np.array([[idx]], dtype=np.int32), (x, ext), label(seed, idx)


# ---------------------------------------------------------------------------------------
# parameters: a state_dict filled from the counter RNG, loadable into the reference modules
# and into ours alike, so fixtures never need to carry weights.
# ---------------------------------------------------------------------------------------
def param(seed: int, name: str, shape) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    key = stream_key(seed, "param:" + name)
    u = uniform01(key, n) * 2.0 - 1.0
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:                      # weight matrices / 1x1 conv: xavier-uniform bound
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:]))
        bound = np.sqrt(6.0 / (fan_in + fan_out))
        v = u * bound
    elif "norm" in name and leaf == "weight":  # LayerNorm gain around 1
        v = 1.0 + 0.1 * u
    elif leaf == "t":                         # GENConv temperature
        v = 1.0 + 0.05 * u
    else:                                     # biases / LayerNorm shifts: small, non-zero
        v = 0.05 * u
    return v.astype(np.float32).reshape(shape)


def state_dict(seed: int, shapes: dict) -> dict:
    """{name: ndarray} for a {name: shape} map (e.g. from `module.state_dict()`)."""
    return {k: param(seed, k, s) for k, s in shapes.items()}
