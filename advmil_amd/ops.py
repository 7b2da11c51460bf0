"""Host-side wrappers over the C ABI (include/advmil_hip.h): raw launches + torch.autograd Functions.

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream, autograd
bookkeeping. Every N-row computation of the AdvMIL path runs in libadvmil_hip.so; there is no
eager fallback -- CPU tensors raise.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import Epilogue

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
# True: backward kernels add parameter gradients straight into the optimizer's flat gradient arena (see _arena_grad)
FUSED_WGRAD = True
# bench.py sets this to a list to bracket every GEMM launch with HIP events on the launch stream:
# entries are (kernel name, (M, N, K, splits), flops, start_event, stop_event)
KERNEL_PROFILE = None
STAMPS = None            # bench.py: a Stamps object -> device wall-clock stamps around the contraction / attention launches (also inside a graph capture)


class Stamps:
    """Measurement aid: `mark` launches advmil_stamp_clock (a one-thread kernel writing the device's constant-rate wall clock) into the
    next slot of a device buffer, in stream order; begin / end marks bracket a launch where it sits in the step -- inside the captured
    graph the stamps are graph nodes, re-written on every replay. `durations_us()` pairs them up after a synchronize."""

    def __init__(self, device, cap=2048):
        self.buf = torch.zeros(cap, dtype=torch.int64, device=device)
        self.tags = []
        self.khz = int(_lib.lib().advmil_clock_rate_khz()) or 100000

    def mark(self, tag):
        i = len(self.tags)
        if i >= self.buf.numel():
            return
        _lib.check(_lib.lib().advmil_stamp_clock(ctypes.c_void_p(self.buf.data_ptr() + 8 * i), _stream()), "stamp_clock")
        self.tags.append(tag)

    def durations_us(self, start=0):
        """[(name, shape, flops, us)] of every bracketed launch from mark `start` on, from the buffer's current contents (synchronize first)."""
        t = self.buf[:len(self.tags)].cpu().tolist()
        out, open_ = [], {}
        for i, (ph, name, shape, flops) in enumerate(self.tags):
            if i < start:
                continue
            if ph == "b":
                open_[(name, shape)] = i
            else:
                j = open_.pop((name, shape), None)
                if j is not None:
                    out.append((name, shape, flops, (t[i] - t[j]) * 1e3 / self.khz))
        return out


def _stamp(ph, name, shape, flops):
    st = STAMPS
    if st is not None:
        st.mark((ph, name, shape, flops))
# no-grad gated-attention passes skip the [N,2D] activations (ADVMIL_FUSED_GATE=0 keeps the two-launch path, for A/B timing)
FUSED_GATE_SCORE = os.environ.get("ADVMIL_FUSED_GATE", "1") != "0"
_ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}


# torch.cuda.current_stream() builds a Stream object per call (2.8 us, tools/probe/host_call_cost.py); an eager step makes ~165
# launches, each of which asks for the stream -> the raw handle straight from the C layer (0.2 us).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"advmil_amd: `{name}` lives on {t.device}; the HIP path needs a ROCm device tensor "
                           "(no CPU fallback in the product path)")
    if t.dtype != torch.float32:
        raise TypeError(f"advmil_amd: `{name}` must be float32, got {t.dtype}")
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def is_bf16_slab(t):
    """A bag / step slab held as ONE bf16 plane (cfg x_storage = 'bf16'): the tensor is its own operand plane, there is no fp32 image."""
    return t is not None and t.dtype == torch.bfloat16


def as_f32(t):
    """fp32 image of a bf16 slab (exact), for the few launches that cannot take it as a plane; kept on the tensor object."""
    if t.dtype != torch.bfloat16:
        return t
    f = t.__dict__.get("_advmil_f32")
    if f is None or f[0] != t._version:
        f = (t._version, t.float())
        t._advmil_f32 = f
    return f[1]


_CONST_ZEROS = {}


def _const_zeros(n, device):
    """A read-only fp32 zero vector (the all-equal scores of the mean pool), kept per (n, device) instead of a fill launch per use.
    Not cached while a graph is being captured: that allocation belongs to the graph's private pool."""
    key = (int(n), str(device))
    t = _CONST_ZEROS.get(key)
    if t is None:
        t = torch.zeros(n, dtype=torch.float32, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _CONST_ZEROS[key] = t
    return t


def _ws(nbytes, device):
    t = torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device=device)
    if _DEFER_KEEP is not None:
        _DEFER_KEEP.append(t)          # a queued merge reads this workspace at the flush: keep the block out of the allocator until then
    return t


# Deferred merges of parameter-gradient partials (include/advmil_hip.h::advmil_defer_sums, csrc/sumq.hip): inside the context every
# accumulating merge launch of the backward kernels (bias / gate / LayerNorm column sums, split-K reduces of the weight gradients) is
# queued and the exit issues them as ONE launch. The handler wraps each of its two backward calls: 14 merge launches per step -> 2.
DEFER_SUMS = os.environ.get("ADVMIL_DEFER_SUMS", "1") != "0"
_DEFER_KEEP = None


class deferred_sums:
    def __enter__(self):
        global _DEFER_KEEP
        self.on = DEFER_SUMS and _DEFER_KEEP is None
        if self.on:
            self.stream = _stream()
            _lib.check(_lib.lib().advmil_defer_sums(self.stream, 1), "defer_sums")
            _DEFER_KEEP = []
        return self

    def __exit__(self, *exc):
        global _DEFER_KEEP
        if self.on:
            keep, _DEFER_KEEP = _DEFER_KEEP, None
            _lib.check(_lib.lib().advmil_defer_sums(self.stream, 0), "flush_sums")
            del keep                     # (same stream as the flush: the allocator's stream order keeps the blocks intact until it has run)
        return False


# ---------------------------------------------------------------------------------------
# counter RNG state (seed in device memory so HIP graphs can replay with fresh randomness)
# ---------------------------------------------------------------------------------------
class IdentityRows:
    """A row layout of the step whose rows draw at their own index (entry of DeviceRng.rows: a single process whose slab carries
    a zero-row pad maps only the stacked layouts; the kernels then skip the per-row lookup)."""
    __slots__ = ("shape",)

    def __init__(self, n):
        self.shape = (int(n),)


class DeviceRng:
    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.seed = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.counter = 0
        self.record = False
        self.log = []
        # Bag-parallel: {layout kind (parallel.rng_row_maps): int64 device tensor of the rows this rank's rows of that layout occupy
        # in the SINGLE-PROCESS step slab}. Every dropout / noise draw is indexed through the map of its call site's layout
        # (SITE_LAYOUTS; rng_row arguments of the C ABI), so the masks do not depend on the world size. None = single process
        # (identity). Set by the handler for the duration of a step's forward and cleared behind it.
        self.rows = None
        self.reset(seed)

    def reset(self, seed):
        """Set the step seed and rewind the call-site counter (not capturable: does a H2D copy)."""
        s = int(seed) & 0xFFFFFFFFFFFFFFFF
        if s >= 1 << 63:
            s -= 1 << 64
        self.seed.fill_(s)
        self.counter = 0
        self.log = []

    def site(self, tag="", shape=None, p=None):
        """A fresh stream id for one dropout/noise call site."""
        self.counter += 1
        if self.record:
            self.log.append((tag, self.counter, shape, p))
        return self.counter

    def advance(self, inc=1):
        """seed += inc on the device (capturable)."""
        _lib.check(_lib.lib().advmil_seed_advance(_p(self.seed), inc, _stream()), "seed_advance")

    def row_map(self, n_rows, tag=None):
        """Bag-parallel row map of the call site `tag` for a tensor of n_rows rows (None in a single process). The site names the
        layouts it can be fed (SITE_LAYOUTS); the one whose map has n_rows rows is taken. Anything else is an error: a silent
        fall-back to local row numbers would draw masks that differ from the single-process run."""
        if self.rows is None:
            return None
        kinds = site_layouts(tag)
        if kinds is None:
            raise RuntimeError(f"bag-parallel: dropout / noise site '{tag}' has no registered row layout (ops.SITE_LAYOUTS); its "
                               "draws would not be world-size invariant")
        for k in kinds:
            m = self.rows.get(k)
            if m is not None and m.shape[0] == int(n_rows):
                return None if isinstance(m, IdentityRows) else m
        raise RuntimeError(f"bag-parallel: site '{tag}' was given {int(n_rows)} rows, none of its layouts {kinds} has that many "
                           f"({ {k: int(v.shape[0]) for k, v in self.rows.items()} })")

    def uniform(self, n, tag="noise", width=None):
        """n uniforms at flat indices 0..n-1 of a fresh stream; `width`: the tensor is [n / width, width] (rows = bags), which
        lets the bag-parallel row map address the single-process rows."""
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        sid = self.site(tag, (n,), None)
        rr = self.row_map(n // width, tag) if width else None
        _lib.check(_lib.lib().advmil_uniform_fill(_p(out), n, _p(self.seed), sid, _p(rr), width or 0, _stream()), "uniform_fill")
        return out


# Row layouts of the step slab a dropout / noise call site can be applied to, by tag prefix (longest prefix wins). The layouts of
# one site never share a row count (L vs 2L, n vs 2n, sum N >= 16 n vs 8 n), so (site, row count) identifies the map even when
# two DIFFERENT layouts of a step happen to have equally many rows (e.g. 2n == sum N / 16 for two 32-patch bags).
SITE_LAYOUTS = {
    "abmil_fc": ("patch",), "gcn_fc": ("patch",), "gcn_phi": ("patch",), "gate_": ("patch", "cluster"),
    "abmil_rho": ("bag",), "misl_fc": ("cluster",), "gapool_": ("region", "region2"), "esat_": ("region",),
    "dx_fc1": ("region", "region2"), "dx_fc2": ("bag", "bag2"), "dy": ("bag", "bag2"), "gen_mlp": ("bag",), "noise": ("bag",),
    "surv_mlp": ("bag",),
}


def site_layouts(tag):
    if not tag:
        return None
    best = None
    for pre, kinds in SITE_LAYOUTS.items():
        if tag.startswith(pre) and (best is None or len(pre) > len(best[0])):
            best = (pre, kinds)
    return None if best is None else best[1]


_RNGS = {}


def default_rng(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _RNGS:
        _RNGS[key] = DeviceRng(device)
    return _RNGS[key]


# ---------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------
_PLAN_CACHE = {}


def set_gemm_mode(mode):
    """'exact' (fp32 MFMA) or 'bf16x3' (split-bf16 on the bf16 matrix pipe, fp32 accumulate) for the fp32 engine."""
    global _MODE_CODE
    code = {"exact": 0, "f32": 0, 0: 0, "bf16x3": 1, "split": 1, 1: 1}[mode]
    _lib.check(_lib.lib().advmil_set_gemm_mode(code), "set_gemm_mode")
    _MODE_CODE = code


_MODE_CODE = None            # mirror of the library's mode (set only through set_gemm_mode / the environment at load time)


def _mode_code():
    global _MODE_CODE
    if _MODE_CODE is None:
        _MODE_CODE = int(_lib.lib().advmil_get_gemm_mode())
    return _MODE_CODE


def get_gemm_mode():
    return "bf16x3" if _mode_code() == 1 else "exact"


def gemm_plan(M, N, K, a_kc=True, b_kc=True):
    """(tile, splits) from the library's launch plan (advmil_gemm_f32_plan_layout); depends on the arithmetic mode."""
    key = (M, N, K, bool(a_kc), bool(b_kc), _mode_code())
    if key not in _PLAN_CACHE:
        t, sp = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().advmil_gemm_f32_plan_layout(1 if a_kc else 0, 1 if b_kc else 0, M, N, K, ctypes.byref(t),
                                                          ctypes.byref(sp)), "gemm_plan")
        _PLAN_CACHE[key] = (t.value, sp.value)
    return _PLAN_CACHE[key]


def gemm_plan_planes(M, N, K, a_kc=True, b_kc=True):
    """Tile code (82 / 83) of the plane-fed LDS-DMA kernel for an NT contraction whose operands both come as Planes, or 0."""
    key = ("pl", M, N, K, bool(a_kc), bool(b_kc), _mode_code())
    if key not in _PLAN_CACHE:
        t = ctypes.c_int(0)
        _lib.check(_lib.lib().advmil_gemm_f32_plan_planes(1 if a_kc else 0, 1 if b_kc else 0, M, N, K, ctypes.byref(t)), "gemm_plan_planes")
        _PLAN_CACHE[key] = t.value
    return _PLAN_CACHE[key]


def gemm_plan_tn_planes(M, N, K):
    """(tile, splits) of the plane-fed TN kernel (both operands [K, .] as Planes: deep-K weight gradients), or (0, 1)."""
    key = ("tn", M, N, K, _mode_code())
    if key not in _PLAN_CACHE:
        t, sp = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().advmil_gemm_f32_plan_tn_planes(M, N, K, ctypes.byref(t), ctypes.byref(sp)), "gemm_plan_tn_planes")
        _PLAN_CACHE[key] = (t.value, sp.value)
    return _PLAN_CACHE[key]


TN_GROUP = os.environ.get("ADVMIL_TN_GROUP", "1") != "0"


def gemm_tn_group(calls):
    """calls: [(A [K, M], B [K, N], out [M, N] view (row pitch = stride(0)), accumulate)], at most 4: out (+)= A^T B for all of them in ONE
    launch (advmil_gemm_tn_group: 64x64 tiles, split-K per member). Small deep-K weight gradients that would each leave most of the chip
    idle."""
    n = len(calls)
    arr = (_lib.GemmTnCall * n)()
    for c, (A, B, out, acc) in zip(arr, calls):
        K, M = A.shape
        c.M, c.N, c.K = M, B.shape[1], K
        c.A, c.lda, c.B, c.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
        c.C, c.ldc, c.accumulate = out.data_ptr(), out.stride(0), 1 if acc else 0
    L = _lib.lib()
    wsb = L.advmil_gemm_tn_group_workspace_bytes(arr, n)
    ws = _ws(wsb, calls[0][0].device) if wsb else None
    _lib.check(L.advmil_gemm_tn_group(arr, n, _p(ws), wsb, _stream()), "gemm_tn_group")


def auto_splits(M, N, K):
    return gemm_plan(M, N, K)[1]


class Planes:
    """bf16x3 operand image of an fp32 matrix: hi = bf16(x), lo = bf16(x - hi), same shape/strides as x
    (include/advmil_hip.h::advmil_epilogue_t.a_hi..c_lo). A contraction given the planes of an operand skips the per-workgroup
    re-split of that operand; results are bit-identical."""
    __slots__ = ("hi", "lo", "fp32_stale")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo               # lo None: a SINGLE-plane operand -- the tensor is bf16 itself (x_storage = "bf16")
        # True: the fp32 tensor these planes belong to was never written (a step slab whose cached bags were staged as planes only,
        # ingest.SlabStager): a contraction given them must read the planes and nothing else
        self.fp32_stale = False

    @property
    def single(self):
        return self.lo is None

    def ptrs(self):
        """OR of the planes' addresses (alignment checks)."""
        return self.hi.data_ptr() | (0 if self.lo is None else self.lo.data_ptr())

    def lo_ptr(self):
        return 0 if self.lo is None else self.lo.data_ptr()

    # The two planes of a tensor live in ONE allocation, lo starting 64 KB behind the end of hi. Measured on the plane-fed two-layer
    # launch (131072 x 512 x 1024, rocprofv3 FETCH_SIZE, tools/pmc_fetch_skew.sh): two separate allocations fetch 1 145 MB per launch
    # for 539 MB of operands -- the hi and the lo stream of the same rows evict each other from L2 before the workgroup that shares
    # the A panel has read them --, one allocation 690-750 MB (64 KB skew: 690), a row-interleaved [hi | lo] layout 778 MB. The
    # launch time does not move (the kernel is not HBM-bound), the traffic does.
    SKEW = 32768             # halfwords

    @staticmethod
    def alloc(shape, device):
        n = 1
        for v in shape:
            n *= int(v)
        pad = (-n) % 8
        buf = torch.empty(2 * n + pad + Planes.SKEW, dtype=torch.bfloat16, device=device)
        return Planes(buf[:n].view(*shape), buf[n + pad + Planes.SKEW:n + pad + Planes.SKEW + n].view(*shape))

    @staticmethod
    def empty_like(x):
        if x.is_contiguous():
            return Planes.alloc(tuple(x.shape), x.device)
        return Planes(torch.empty_strided(x.shape, x.stride(), dtype=torch.bfloat16, device=x.device),
                      torch.empty_strided(x.shape, x.stride(), dtype=torch.bfloat16, device=x.device))

    def view_rows(self, r0, r1):
        return Planes(self.hi[r0:r1], None if self.lo is None else self.lo[r0:r1])


# Operand planes of tensors that several contractions read (bf16x3 mode). They always travel as an ATTRIBUTE of the tensor object
# they describe (never keyed by address: the allocator reuses addresses), so they die with it:
#   * a step slab X: attribute set by the handler (MyHandler._slab_planes), which keeps the planes of a resident slab on the first
#     bag's tensor object for as long as its storage is unchanged (version counter);
#   * an activation emitted with its planes by the producing contraction's epilogue: attribute `_advmil_planes` on the result;
#   * a weight living in a FlatAdam arena: attribute `_advmil_planes` on the parameter, kept current by the Adam kernel and
#     re-derived when torch writes the parameter (its version counter moves: load_state_dict, copy_).
USE_PLANES = os.environ.get("ADVMIL_PLANES", "1") != "0"
# Measured same-box (tools/ab_bench.sh, 16 x 8k ABMIL step): planes for the slab, the weights and the eval-pass h pay (+2.6 %: the
# embedding FCs and the fused gate score run on the plane-fed kernel); emitting planes of dG to run dh as an NT plane contraction
# costs 2 % (the extra 403 MB of plane writes outweigh the staging they save) -> off unless asked for. Planes of the
# memo-replayed (dropped) h were neutral while the training-pass gate contraction took the 256x192 tile with the old epilogue; with
# the persistent 256x256 tile and the plain streaming epilogue they pay (+0.4-0.8 %, two same-box A/B runs) -> on.
# round 6: dh WITH the first layer's activation backward in its epilogue (rank-1 term + bit mask + bias column sums) on the plane-fed NT
# kernel (B = the planes of Wab^T: one small transposing copy), instead of the generic kernel's 256x192 tile
DH_NT_FUSED = os.environ.get("ADVMIL_DH_NT_FUSED", "1") != "0"
# the first layer's activation / dropout backward in the epilogue of the pool's dh contraction (rank-1 term + mask + bias column sums)
ACT_BWD_IN_DH = os.environ.get("ADVMIL_ACT_BWD_IN_DH", "1") != "0"
# dG of the gate backward as planes ONLY (no fp32 copy): its two consumers take the A operand pre-split (-30 us each, no extra bytes)
DG_PLANES_ONLY = os.environ.get("ADVMIL_DG_PLANES_ONLY", "1") != "0"
# weight gradients dY^T X of the layers applied to the slab: X's planes (already resident for the forward) feed the B operand
DW_PLANES = os.environ.get("ADVMIL_DW_PLANES", "1") != "0"
MEMO_PLANES = os.environ.get("ADVMIL_MEMO_PLANES", "1") != "0"
# round 6: the generator's first layer over the step slab (the two-layer launch) leaves its [rows, hid] output as operand planes ONLY -- the
# gate contraction reads the planes anyway, the pooling kernels and the dropout replay of the memoized output read them too (h = hi + lo,
# 2^-17 relative, the arithmetic of every bf16x3 contraction): 201 MB less written by the launch and by the replay, and the pooling pass
# no longer pays for the write-back of those dirty lines (tools/probe/pool_instep.py). 0 = fp32 rows beside the planes, as until round 5.
H_PLANES_ONLY = os.environ.get("ADVMIL_H_PLANES_ONLY", "1") != "0"
# round 6: the TRAINING pass's gate score in the gate contraction's epilogue (branches in pair blocks of 32 columns, keep bits drawn by the
# dropout pass over the memoized first layer): gate_score_kernel's pass over the stored [rows, 2D] activations does not run
FUSED_GATE_TRAIN = os.environ.get("ADVMIL_FUSED_GATE_TRAIN", "1") != "0"
# operand planes for EVERY step slab of >= 4096 rows, not only for those that fill the chip with the plane-fed NT kernel's 256-row tiles: the
# deep-K weight gradients (plane-fed TN kernel from K = 8192), the planes-only dpre / dG hand-overs and the dh epilogue fusion then also
# apply to the 1-4 bag steps of a strong split
SLAB_PLANES_ANY = os.environ.get("ADVMIL_SLAB_PLANES_ANY", "1") != "0"
GENERIC_PLANES = os.environ.get("ADVMIL_GENERIC_PLANES", "1") != "0"
# [B <= 32, d] linear layers on the fp32-FMA kernels (csrc/optim.hip small_linear_*) instead of the 64x64-tile MFMA contraction
SMALL_LINEAR = os.environ.get("ADVMIL_SMALL_LINEAR", "1") != "0"
# ... for up to this many rows. In-graph, forward + backward of a [B, 384] -> 384 layer (tools/probe/small_linear_time.py): B = 1-2:
# 19-20 us against 32 us; B = 16: 39 against 39 us; B = 32: 35 against 27 us (per row the FMA kernels do 64 x the work per lane)
SMALL_LINEAR_ROWS = int(os.environ.get("ADVMIL_SMALL_LINEAR_ROWS", "8"))


def slab_takes_planes(rows, C):
    """Will a step slab of `rows` x C fp32 rows be read through its operand planes (so that staging may skip its fp32 rows)? The ONE
    rule of ingest.SlabStager.ready() -- which back-fills the fp32 rows when it says no -- and MyHandler._slab_build_static, which
    attaches the planes when it says yes."""
    return bool(rows >= 4096 and USE_PLANES and get_gemm_mode() == "bf16x3" and (SLAB_PLANES_ANY or gemm_plan_planes(rows, 128, C)))


def planes_transposed(pl):
    """Planes of x^T from the planes of a 2-D x: ONE transposing copy when hi and lo live in one allocation (the optimizer's plane arena,
    Planes.alloc), else two."""
    R, C = pl.hi.shape
    out = Planes.alloc((C, R), pl.hi.device)
    d = (pl.lo.data_ptr() - pl.hi.data_ptr()) // 2
    do = (out.lo.data_ptr() - out.hi.data_ptr()) // 2
    if (pl.hi.untyped_storage().data_ptr() == pl.lo.untyped_storage().data_ptr() and d > 0 and pl.hi.stride() == pl.lo.stride()
            and pl.hi.stride(1) == 1):
        src = pl.hi.as_strided((2, C, R), (d, 1, pl.hi.stride(0)), pl.hi.storage_offset())
        dst = out.hi.as_strided((2, C, R), (do, R, 1), out.hi.storage_offset())
        dst.copy_(src)
    else:
        out.hi.copy_(pl.hi.t())
        out.lo.copy_(pl.lo.t())
    return out


def _token(M, N, device):
    """An fp32 [M, N] tensor that carries shape and autograd identity of a result whose VALUES exist as operand planes only: never
    written, never to be read (NaN-filled under ADVMIL_POISON_TOKENS, see PlaneHandover)."""
    t = torch.empty(M, N, dtype=torch.float32, device=device)
    if POISON_TOKENS:
        t.fill_(float("nan"))
    return t


def planes_of(x):
    """Planes of an activation / slab tensor, or None. A bf16 slab is its own (single) plane."""
    if is_bf16_slab(x):
        return Planes(x, None)
    return getattr(x, "_advmil_planes", None) if USE_PLANES else None


def weight_planes(W):
    """Planes of a parameter held in a FlatAdam arena ([N, K] view), or None."""
    if not USE_PLANES:
        return None
    ent = getattr(W, "_advmil_planes", None)
    if ent is None:
        return None
    owner, ver, pl = ent
    if ver != W._version:                    # torch wrote the parameter since the planes were derived: refresh the whole arena
        owner.refresh_planes()
        owner, ver, pl = W._advmil_planes
    return pl


def split_planes(x, out=None):
    """x (fp32, dense storage) -> Planes(hi, lo)."""
    _chk(x, "x")
    if out is None:
        out = Planes.empty_like(x)
    _lib.check(_lib.lib().advmil_split_planes(_p(x), x.numel(), _p(out.hi), _p(out.lo), _stream()), "split_planes")
    return out


def pre_a_tile_ok(tile, a_kc, b_kc, b_planes=False):
    """Is the contraction kernel of this tile built for an A operand that arrives as planes only (csrc/gemm_f32.hip dispatch)?"""
    return tile in (22, 12, 11) or (tile == 43 and not b_kc and not b_planes) or (tile in (34, 24) and not a_kc and not b_kc and b_planes)


def gemm(A, B, a_kc, b_kc, M, N, K, out=None, ldc=None, bias=None, act0=0, act1=None, act_split=None, drop_p=0.0,
         seed=None, stream_id=0, rowv=None, colv=None, rowseg=None, maskref=None, mask_scale=1.0, accumulate=False,
         alpha=1.0, splits=None, tile=0, a_planes=None, b_planes=None, c_planes=None, c_planes_only=False, gate_wc=None, rng_row=None,
         colsum=None, maskbits=None, gate_bits=None, c_rows_pair32=False):
    """C[M,N] = epilogue(alpha * op(A) op(B)); see include/advmil_hip.h::advmil_gemm_f32. a_planes / b_planes: optional
    Planes of A / B; c_planes: Planes to receive the split of the final C (pitch ldc)."""
    planes_only_a = A is None
    if planes_only_a:
        # A exists as planes only (dG of the gate backward): the launch must land on a kernel instantiated for a pre-split A operand
        if a_planes is None or get_gemm_mode() != "bf16x3":
            raise ValueError("gemm(A=None) needs a_planes in bf16x3 mode")
        A = a_planes.hi                       # (pointer and pitch only: never read as fp32)
    else:
        if is_bf16_slab(A):                   # a bf16 slab: it IS its hi plane; its pointer only carries the pitch
            a_planes = a_planes or Planes(A, None)
            planes_only_a = True
        else:
            _chk(A, "A")
            if a_planes is not None and a_planes.fp32_stale:
                planes_only_a = True          # (A's fp32 rows were never written: pointer and pitch only)
    b_single = False
    planes_only_b = B is None                 # B exists as (two) planes only (Wab^T of the pooling backward's dh): plane-fed kernels only
    if planes_only_b:
        if b_planes is None or b_planes.single or get_gemm_mode() != "bf16x3" or not (82 <= tile <= 86 or 91 <= tile <= 93):
            raise ValueError("gemm(B=None) needs two-plane b_planes, bf16x3 mode and a plane-fed tile")
        B = b_planes.hi                       # (pointer and pitch only: never read as fp32)
    elif is_bf16_slab(B):
        b_planes = b_planes or Planes(B, None)
        b_single = True
    else:
        _chk(B, "B")
    b_only = b_single or (b_planes is not None and b_planes.fp32_stale)       # B must be read from its plane(s)
    if (planes_only_a or b_only) and get_gemm_mode() != "bf16x3":
        raise ValueError("bf16 slabs need gemm_mode 'bf16x3' (ops.as_f32 gives their fp32 image)")
    if (tile == 0 and a_planes is not None and b_planes is not None and (splits is None or splits == 1) and not b_planes.single
            and not ((a_planes.ptrs() | b_planes.ptrs()) & 15)
            and a_planes.hi.stride(0) % 8 == 0 and b_planes.hi.stride(0) % 8 == 0):
        ptile = gemm_plan_planes(M, N, K, a_kc, b_kc)          # both operands pre-split: the plane-fed LDS-DMA kernel, if the shape fits
        if ptile:
            tile, splits = ptile, 1
            if gate_wc is not None and gate_bits is not None:
                tile = 85 if (N % 256 == 0 and (M // 256) * (N // 256) >= 256) else 86      # training form: a plain tile, C written too
            elif gate_wc is not None and N % 256 == 0 and (M // 256) * (N // 256) >= 384:
                tile = 84                                      # the fused gate score's own 256x256 form
            elif (gate_wc is None and N % 256 == 0 and (M // 256) * (N // 256) >= 384 and rowv is None and maskref is None
                  and not accumulate and (drop_p <= 0.0 or seed is None)):
                tile = 85                                      # plain bias + activation (+ planes): the 256x256 form (9 % over 256x192)
    if (tile == 0 and splits is None and a_planes is not None and b_planes is not None and not a_kc and not b_kc and gate_wc is None
            and not a_planes.single and not ((a_planes.ptrs() | b_planes.ptrs()) & 15)
            and a_planes.hi.stride(0) % 8 == 0 and b_planes.hi.stride(0) % 8 == 0):
        ttile, tsplits = gemm_plan_tn_planes(M, N, K)          # both [K, .] operands pre-split: the plane-fed TN kernel, if the shape fits
        if ttile:
            tile, splits = ttile, tsplits
    if gate_wc is not None:
        # fused gate score (advmil_epilogue_t.gate_wc): B / bias hold the INTERLEAVED branches; returns per-row partial scores
        # [M, column blocks] instead of C
        tile, _ = gemm_plan(M, N, K, a_kc, b_kc) if tile == 0 else (tile, 1)
        npart = _lib.lib().advmil_gemm_f32_gate_blocks(tile, N)
        gate_out = torch.empty(M, npart, dtype=torch.float32, device=A.device)
        splits, ldc = 1, N
        if gate_bits is None:
            out = None
        else:             # training form (advmil_epilogue_t.gate_bits_a): the activations are stored as well, in pair-block column order
            if not (85 <= tile <= 86):
                raise ValueError("gemm(gate_bits=...) needs the plane-fed plain tiles (operands as planes, slab-sized M)")
            out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    elif c_planes_only:                       # the result is consumed as a bf16x3 operand only: its planes are written, no fp32 C at all
        if c_planes is None or out is not None or accumulate:
            raise ValueError("gemm(c_planes_only=True) needs c_planes, no `out`, no accumulate")
        splits, ldc = 1, N
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
        ldc = N
    elif ldc is None:
        ldc = out.stride(0)
    lda = A.stride(0)
    ldb = B.stride(0)
    e = Epilogue()                            # zero-initialised: only the fields that differ are written (host time per launch)
    if bias is not None:
        e.bias = bias.data_ptr()
    e.act0 = act0
    e.act1 = act0 if act1 is None else act1
    e.act_split = (1 << 30) if act_split is None else act_split
    use_seed = seed is not None and drop_p > 0.0
    if use_seed:
        e.drop_p = float(drop_p)
        e.seed = seed.data_ptr()
        e.stream_id = stream_id
        if rng_row is not None:
            e.rng_row = rng_row.data_ptr()
    elif drop_p:
        e.drop_p = float(drop_p)
    if rowv is not None:
        e.rowv = rowv.data_ptr()
    if colv is not None:
        e.colv = colv.data_ptr()
    if rowseg is not None:
        e.rowseg = rowseg.data_ptr()
    if maskref is not None:
        e.maskref = maskref.data_ptr()
        e.ldmask = maskref.stride(0)
    e.mask_scale = float(mask_scale)
    if accumulate:
        e.accumulate = 1
    e.alpha = float(alpha)
    if a_planes is not None:
        e.a_hi, e.a_lo = a_planes.hi.data_ptr(), a_planes.lo_ptr()
    if b_planes is not None:
        e.b_hi, e.b_lo = b_planes.hi.data_ptr(), b_planes.lo_ptr()
    if c_planes is not None:
        e.c_hi, e.c_lo = c_planes.hi.data_ptr(), c_planes.lo.data_ptr()
    if gate_wc is not None:
        e.gate_wc, e.gate_out, e.gate_np = gate_wc.data_ptr(), gate_out.data_ptr(), npart
        if gate_bits is not None:
            e.gate_bits_a, e.gate_bits_b, e.ldgbits = gate_bits[0].data_ptr(), gate_bits[1].data_ptr(), gate_bits[0].stride(0)
    if c_rows_pair32:
        e.c_rows_pair32 = 1
    if colsum is not None:
        e.colsum = colsum.data_ptr()
    if maskbits is not None:
        e.maskbits, e.ldbits = maskbits.data_ptr(), maskbits.stride(0)
        e.mask_scale = float(mask_scale)
    if splits is None:
        ptile, splits = gemm_plan(M, N, K, a_kc, b_kc)
        if tile == 0:
            tile = ptile
    if accumulate and splits > 1 and _DEFER_KEEP is not None and out is not None and not _is_arena(out):
        # Inside ops.deferred_sums() the `C += partials` fold of a split-K launch is QUEUED until the context's exit (advmil_defer_sums):
        # right for the optimizer's gradient arena, which nothing reads before the step, wrong for any other destination -- the next
        # launch would read C without the partials. Such a call takes the fold-free plan.
        splits = 1
    if b_only and not (91 <= tile <= 93) and tile not in (22, 12, 11) and not (tile in (34, 24) and not a_kc and not b_kc):
        tile = 22                             # a bf16 / planes-only B operand has no fp32 image: land on a kernel that stages B from its plane(s)
    if planes_only_a and a_planes.single and not (82 <= tile <= 86) and tile not in (22, 12, 11):
        tile = 22                             # (single-plane A: only these forms are built for it)
    if planes_only_a and a_planes.fp32_stale and not (82 <= tile <= 86) and not (91 <= tile <= 93) and tile not in (22, 12, 11) \
            and not pre_a_tile_ok(tile, a_kc, b_kc, b_planes is not None):
        tile = 22                             # (a slab staged as planes only: land on a form that stages A from its planes)
    if planes_only_a and not (82 <= tile <= 86) and not (91 <= tile <= 93) and not pre_a_tile_ok(tile if tile else gemm_plan(M, N, K, a_kc, b_kc)[0], a_kc, b_kc,
                                                                     b_planes is not None):       # (82-85: plane-fed, reads planes only)
        raise ValueError(f"gemm(A=None): tile {tile} has no pre-split-A instantiation for this layout")
    L = _lib.lib()
    wsb = L.advmil_gemm_f32_workspace_bytes(M, N, splits)
    ws = _ws(wsb, A.device) if wsb else None
    prof = KERNEL_PROFILE
    name = None
    if prof is not None or STAMPS is not None:
        name = ("gemm_nt_planes_kernel<%d>" % (tile - 80)) if 82 <= tile <= 84 else "gemm_nt_planes_kernel<4,plain>" if tile == 85 else "gemm_nt_planes_kernel<2,plain>" if tile == 86 else \
            ("gemm_tn_planes_kernel<%d>" % tile) if 91 <= tile <= 93 else \
            "gemm_f32_kernel<%d,%d,%d,%d>" % (bool(a_kc), bool(b_kc), tile // 10, tile % 10)
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if STAMPS is not None and M * N * K >= (1 << 30):
        _stamp("b", name, (M, N, K, splits), 2.0 * M * N * K)
    _lib.check(L.advmil_gemm_f32_tiled(1 if a_kc else 0, 1 if b_kc else 0, M, N, K, _p(A), lda, _p(B), ldb, _p(out), ldc,
                                       ctypes.byref(e), splits, tile, _p(ws), wsb, _stream()), f"gemm_f32[{M}x{N}x{K}]")
    if STAMPS is not None and M * N * K >= (1 << 30):
        _stamp("e", name, (M, N, K, splits), 2.0 * M * N * K)
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((name, (M, N, K, splits), 2.0 * M * N * K, e0, e1))
    if gate_wc is not None and gate_bits is not None:
        return out, gate_out
    return gate_out if gate_wc is not None else out


# the ESAT in-projection writes q | k | v as operand planes only (no fp32 qkv, no split pass in front of the attention kernels)
ATTN_QKV_PLANES = os.environ.get("ADVMIL_ATTN_QKV_PLANES", "1") != "0"
TWO_LAYERS_MIN_TILES = int(os.environ.get("ADVMIL_TWO_LAYERS_MIN_TILES", "256"))
TWO_LAYERS_NARROW = os.environ.get("ADVMIL_TWO_LAYERS_NARROW", "1") != "0"


def gemm_two_layers_tile(M, N1, N2, K):
    """Tile code of the ONE plane-fed launch that runs act(x W1^T + b1) and act(x W2^T + b2) over the same rows -- 85: persistent 256x256
    tiles; 86: 256x128 tiles, for a slab too short to give every CU a 256x256 tile (the 16384 rows of a 2-bag step: 128 against 256 tiles)
    -- or 0 when the shapes do not qualify."""
    if not (get_gemm_mode() == "bf16x3" and USE_PLANES and M >= 4096 and M % 256 == 0 and K % 32 == 0 and K >= 64 and N1 % 32 == 0
            and M * K * 2 < (1 << 32)):
        return 0
    if (N1 + N2) % 256 == 0 and (M // 256) * ((N1 + N2) // 256) >= TWO_LAYERS_MIN_TILES:
        return 85
    if TWO_LAYERS_NARROW and (N1 + N2) % 128 == 0 and N1 % 128 == 0 and (M // 256) * ((N1 + N2) // 128) >= TWO_LAYERS_MIN_TILES:
        return 86
    return 0


def gemm_two_layers_ok(M, N1, N2, K):
    return gemm_two_layers_tile(M, N1, N2, K) != 0


def gemm_two_layers(x, xpl, W1, w1pl, b1, act1, W2, w2pl, b2, act2, emit_planes1=False, y1_planes_only=False):
    """(y1, y2, planes of y1 | None): y1 = act1(x W1^T + b1) [M, N1], y2 = act2(x W2^T + b2) [M, N2] from ONE launch that stages
    every row of x once (advmil_epilogue_t two-layer form). All operands as Planes; shapes checked by gemm_two_layers_ok.
    The two weight matrices' planes are stacked for the launch (two ~1 MB copies)."""
    M, K = x.shape
    N1, N2 = W1.shape[0], W2.shape[0]

    def pair(pl, N):
        """[2, N, K] view over a weight's hi and lo plane when both live in one allocation (the optimizer's plane arena), else None."""
        hi, lo = pl.hi.reshape(N, K), pl.lo.reshape(N, K)
        d = (lo.data_ptr() - hi.data_ptr()) // 2
        if hi.untyped_storage().data_ptr() != lo.untyped_storage().data_ptr() or d <= 0 or not hi.is_contiguous() or not lo.is_contiguous():
            return None
        return hi.as_strided((2, N, K), (d, K, 1), hi.storage_offset())
    v1, v2 = pair(w1pl, N1), pair(w2pl, N2)
    if v1 is not None and v2 is not None:     # ONE stacking launch for both planes of both layers
        stk = torch.cat((v1, v2), dim=1)
        wcat = Planes(stk[0], stk[1])
    else:
        wcat = Planes(torch.cat((w1pl.hi.reshape(N1, K), w2pl.hi.reshape(N2, K)), dim=0), torch.cat((w1pl.lo.reshape(N1, K), w2pl.lo.reshape(N2, K)), dim=0))
    dev = x.device
    # y1_planes_only: layer 1's output exists as operand planes only (y1 is an unwritten token)
    y1 = _token(M, N1, dev) if y1_planes_only else torch.empty(M, N1, dtype=torch.float32, device=dev)
    y2 = torch.empty(M, N2, dtype=torch.float32, device=dev)
    cpl = None
    if emit_planes1 or y1_planes_only:
        cpl = Planes.alloc((M, N1), dev)
    e = Epilogue()
    e.bias = None if b1 is None else b1.data_ptr()
    e.bias2 = None if b2 is None else b2.data_ptr()
    e.act0, e.act1, e.act_split = act1, act2, N1
    e.alpha = 1.0
    e.a_hi, e.a_lo = xpl.hi.data_ptr(), xpl.lo_ptr()
    e.b_hi, e.b_lo = wcat.hi.data_ptr(), wcat.lo.data_ptr()
    if cpl is not None:
        e.c_hi, e.c_lo = cpl.hi.data_ptr(), cpl.lo.data_ptr()
    e.c2, e.ldc2, e.n_split = y2.data_ptr(), N2, N1
    prof = KERNEL_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _stamp("b", f"gemm_nt_planes_kernel<4,two layers,{N1}>", (M, N1 + N2, K, 1), 2.0 * M * (N1 + N2) * K)
    _lib.check(_lib.lib().advmil_gemm_f32_tiled(1, 1, M, N1 + N2, K, _p(x), x.stride(0), _p(W1), K, None if y1_planes_only else _p(y1), N1,
                                                ctypes.byref(e), 1, gemm_two_layers_tile(M, N1, N2, K), None, 0, _stream()),
               f"gemm_two_layers[{M}x({N1}+{N2})x{K}]")
    if cpl is not None:
        cpl.fp32_stale = bool(y1_planes_only)
    _stamp("e", f"gemm_nt_planes_kernel<4,two layers,{N1}>", (M, N1 + N2, K, 1), 2.0 * M * (N1 + N2) * K)
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((f"gemm_nt_planes_kernel<4,two layers,{N1}>", (M, N1 + N2, K, 1), 2.0 * M * (N1 + N2) * K, e0, e1))
    return y1, y2, cpl


def gate_score(ab, wc, bc, N, D, p=0.0, seed=None, stream_a=0, stream_b=0, rng_row=None):
    s = torch.empty(N, dtype=torch.float32, device=ab.device)
    sd = seed if p > 0.0 else None
    _lib.check(_lib.lib().advmil_gate_score_fwd(_p(ab), _p(wc), _p(bc), p, _p(sd), stream_a, stream_b, N, D, _p(s),
                                                _p(rng_row if sd is not None else None), _stream()), "gate_score_fwd")
    return s


class Segments:
    """Row partition of a step slab: bag b owns rows [ptr[b], ptr[b+1]). Holds the device arrays the segmented kernels read.
    Both arrays are assembled on the host in ONE pinned buffer and sent with ONE asynchronous copy (no stream sync); construct
    it OUTSIDE HIP-graph capture."""

    def __init__(self, lens, device):
        import numpy as np
        self.lens = [int(v) for v in lens]
        self.nseg = len(self.lens)
        self.total = sum(self.lens)
        self.max_len = max(self.lens)
        self.device = torch.device(device)
        offs = [0]
        for v in self.lens:
            offs.append(offs[-1] + v)
        self.offsets = offs
        nb = (8 * (self.nseg + 1) + 15) // 16 * 16
        host = torch.empty(nb + 4 * self.total, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        hv = host.numpy()
        hv[:8 * (self.nseg + 1)].view(np.int64)[:] = offs
        hv[nb:].view(np.int32)[:] = np.repeat(np.arange(self.nseg, dtype=np.int32), self.lens)
        buf = host.to(self.device, non_blocking=True)
        self._host, self._buf = host, buf
        self.ptr = buf[:8 * (self.nseg + 1)].view(torch.int64)
        self.rowseg = buf[nb:].view(torch.int32)
        self._div = {}

    @property
    def rowseg_long(self):
        if "long" not in self._div:
            self._div["long"] = self.rowseg.to(torch.long)
        return self._div["long"]

    def div(self, k):
        """Segments of the k-fold pooled rows (regions of 16 patches)."""
        if k not in self._div:
            assert all(v % k == 0 for v in self.lens)
            self._div[k] = Segments([v // k for v in self.lens], self.device)
        return self._div[k]

    def twice(self):
        """The partition of two copies of the slab stacked on top of each other (2 * nseg segments)."""
        if "x2" not in self._div:
            self._div["x2"] = Segments(self.lens + self.lens, self.device)
        return self._div["x2"]

    def uniform(self, rows):
        """nseg segments of `rows` rows each (e.g. the 8 cluster rows of every bag)."""
        key = ("u", rows)
        if key not in self._div:
            self._div[key] = Segments([rows] * self.nseg, self.device)
        return self._div[key]


def softmax_pool(s, h, N, D, seg=None, hpl=None):
    """hpl: h held as its operand planes (two-plane Planes, row pitch of h): the pooling reads hi + lo instead of fp32 rows (h itself may
    be an unwritten token then)."""
    L = _lib.lib()
    nseg = 1 if seg is None else seg.nseg
    mlen = N if seg is None else seg.max_len
    A = torch.empty(N, dtype=torch.float32, device=h.device)
    pooled = torch.empty(nseg, D, dtype=torch.float32, device=h.device)
    wsb = L.advmil_softmax_pool_workspace_bytes(mlen, D, nseg)
    ws = _ws(wsb, h.device)
    big = STAMPS is not None and N * D >= (1 << 24)           # bench.py: the call timed where it sits in the step (`flops` slot = bytes)
    if big:
        _stamp("b", "softmax_pool_fwd", (N, D, nseg), 4.0 * N * D + 12.0 * N)
    if hpl is not None:
        _lib.check(L.advmil_softmax_pool_fwd_planes(_p(s), _p(hpl.hi), _p(hpl.lo), hpl.hi.stride(0), N, D, nseg,
                                                    _p(None if seg is None else seg.ptr), mlen, _p(A), _p(pooled), _p(ws), wsb, _stream()),
                   "softmax_pool_fwd_planes")
    else:
        _lib.check(L.advmil_softmax_pool_fwd(_p(s), _p(h), h.stride(0), N, D, nseg, _p(None if seg is None else seg.ptr), mlen, _p(A),
                                             _p(pooled), _p(ws), wsb, _stream()), "softmax_pool_fwd")
    if big:
        _stamp("e", "softmax_pool_fwd", (N, D, nseg), 4.0 * N * D + 12.0 * N)
    return A, pooled


def softmax_pool_mean(s, h, N, D, seg=None):
    """(A, pooled, mean): softmax_pool plus the per-bag unweighted mean of h's rows from the same pass (D % 8 == 0)."""
    L = _lib.lib()
    nseg = 1 if seg is None else seg.nseg
    mlen = N if seg is None else seg.max_len
    A = torch.empty(N, dtype=torch.float32, device=h.device)
    pooled = torch.empty(nseg, D, dtype=torch.float32, device=h.device)
    mean = torch.empty(nseg, D, dtype=torch.float32, device=h.device)
    wsb = L.advmil_softmax_pool_mean_workspace_bytes(mlen, D, nseg)
    ws = _ws(wsb, h.device)
    _lib.check(L.advmil_softmax_pool_mean_fwd(_p(s), _p(h), h.stride(0), N, D, nseg, _p(None if seg is None else seg.ptr), mlen, _p(A),
                                              _p(pooled), _p(mean), _p(ws), wsb, _stream()), "softmax_pool_mean_fwd")
    return A, pooled, mean


def softmax_pool_bwd(dpooled, dA, A, h, N, D, seg=None, hpl=None):
    L = _lib.lib()
    nseg = 1 if seg is None else seg.nseg
    mlen = N if seg is None else seg.max_len
    ds = torch.empty(N, dtype=torch.float32, device=h.device)
    wsb = L.advmil_softmax_pool_workspace_bytes(mlen, D, nseg)
    ws = _ws(wsb, h.device)
    if hpl is not None:
        _lib.check(L.advmil_softmax_pool_bwd_planes(_p(dpooled), _p(dA), _p(A), _p(hpl.hi), _p(hpl.lo), hpl.hi.stride(0), N, D, nseg,
                                                    _p(None if seg is None else seg.ptr), mlen, _p(ds), _p(ws), wsb, _stream()),
                   "softmax_pool_bwd_planes")
    else:
        _lib.check(L.advmil_softmax_pool_bwd(_p(dpooled), _p(dA), _p(A), _p(h), h.stride(0), N, D, nseg,
                                             _p(None if seg is None else seg.ptr), mlen, _p(ds), _p(ws), wsb, _stream()),
                   "softmax_pool_bwd")
    return ds


def dropout_planes(pl, M, N, p, seed, sid, rng_row=None, gate=None):
    """(Planes of dropout(x), keep-and-positive bits [M, N / 32]) from the Planes of x [M, N]: the train-mode forward of a layer whose
    eval-mode output is held as planes only (include/advmil_hip.h::advmil_dropout_planes). The result is planes-only too.
    gate = (p_gate, stream a, stream b): also the keep bits of the gated attention scorer's two branch dropouts -> (out, bits, (bits_a, bits_b))."""
    out = Planes.alloc((M, N), pl.hi.device)
    out.fp32_stale = True
    bits = torch.empty(M, N // 32, dtype=torch.int32, device=pl.hi.device)
    gb = None
    if gate is not None:
        gb = (torch.empty(M, N // 32, dtype=torch.int32, device=pl.hi.device), torch.empty(M, N // 32, dtype=torch.int32, device=pl.hi.device))
    _lib.check(_lib.lib().advmil_dropout_planes(_p(pl.hi), _p(pl.lo), M, N, float(p), _p(seed), sid, _p(rng_row), _p(out.hi), _p(out.lo),
                                                _p(bits), 0.0 if gate is None else float(gate[0]), 0 if gate is None else int(gate[1]),
                                                0 if gate is None else int(gate[2]), _p(None if gb is None else gb[0]),
                                                _p(None if gb is None else gb[1]), _stream()), "dropout_planes")
    return (out, bits, gb) if gate is not None else (out, bits)


def planes_f32(pl):
    """fp32 image hi + lo of a planes-only tensor (the escape hatch of the rare consumers that read rows: two elementwise launches)."""
    return pl.hi.float().add_(pl.lo)


def gate_bwd(ab, ds, wc, N, D, p=0.0, seed=None, stream_a=0, stream_b=0, dwc=None, dbc=None, dbias=None, rng_row=None, planes=None,
             planes_only=False, pair32=False):
    """dwc/dbc/dbias given -> gradients are ADDED into them (views of the gradient arena). planes_only: dG is written as its
    bf16x3 operand planes alone (returned dG is None)."""
    L = _lib.lib()
    dev = ab.device
    acc = dwc is not None
    dG = None if (planes_only and planes is not None) else torch.empty(N, 2 * D, dtype=torch.float32, device=dev)
    if not acc:
        dwc = torch.empty(D, dtype=torch.float32, device=dev)
        dbc = torch.empty(1, dtype=torch.float32, device=dev)
        dbias = torch.empty(2 * D, dtype=torch.float32, device=dev)
    wsb = L.advmil_gate_bwd_workspace_bytes(N, D)
    ws = _ws(wsb, dev)
    sd = seed if p > 0.0 else None
    _lib.check(L.advmil_gate_bwd(_p(ab), _p(ds), _p(wc), p, _p(sd), stream_a, stream_b, N, D, _p(dG), _p(dwc), _p(dbc),
                                 _p(dbias), 1 if acc else 0, _p(rng_row if sd is not None else None),
                                 _p(None if planes is None else planes.hi), _p(None if planes is None else planes.lo), 1 if pair32 else 0,
                                 _p(ws), wsb, _stream()), "gate_bwd")
    return dG, dwc, dbc, dbias


def act_dropout_bwd(dy, y, act, M, N, p=0.0, seed=None, stream_id=0, want_bias=True, db_out=None, rng_row=None, planes=None,
                    planes_only=False, bits=None):
    L = _lib.lib()
    dpre = None if (planes_only and planes is not None) else torch.empty(M, N, dtype=torch.float32, device=dy.device)
    acc = db_out is not None
    db = db_out if acc else (torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None)
    need = acc or want_bias
    wsb = L.advmil_colsum_workspace_bytes(M, N) if need else 0
    ws = _ws(wsb, dy.device) if need else None
    sd = seed if p > 0.0 else None
    _lib.check(L.advmil_act_dropout_bwd(_p(dy), _p(y), act, p, _p(sd), stream_id, M, N, _p(dpre), _p(db), 1 if acc else 0,
                                        _p(rng_row if sd is not None else None), _p(None if planes is None else planes.hi),
                                        _p(None if planes is None else planes.lo), _p(bits), _p(ws), wsb, _stream()), "act_dropout_bwd")
    return dpre, db


def colsum(x, M, N, out=None):
    L = _lib.lib()
    acc = out is not None
    if not acc:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    wsb = L.advmil_colsum_workspace_bytes(M, N)
    ws = _ws(wsb, x.device)
    _lib.check(L.advmil_colsum(_p(x), M, N, _p(out), 1 if acc else 0, _p(ws), wsb, _stream()), "colsum")
    return out


# (Region-level ESAT activations leaving their producers WITH operand planes -- ADVMIL_ROW_PLANES, rounds 4-5 -- measured net slower twice:
# emitting the planes costs what the plane-fed kernel saves on 32768-row layers. Retired in round 6; profiles/r05_ab_log.txt.)


def ln_relu_mean16_fwd(y, gamma, beta, N, d, eps=1e-5, planes=None, dup=1):
    dev = y.device
    emb = torch.empty(dup * (N // 16), d, dtype=torch.float32, device=dev)
    mean = torch.empty(N, dtype=torch.float32, device=dev)
    rstd = torch.empty(N, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().advmil_ln_relu_mean16_fwd(_p(y), _p(gamma), _p(beta), eps, N, d, _p(emb), _p(mean), _p(rstd),
                                                    _p(None if planes is None else planes.hi), _p(None if planes is None else planes.lo),
                                                    int(dup), _stream()), "ln_relu_mean16_fwd")
    return emb, mean, rstd


def ln_relu_mean16_bwd(demb, y, gamma, beta, mean, rstd, N, d, dg_out=None, db_out=None, ycol_out=None, planes=None, dup=1):
    """ycol_out: optional [d] accumulator that receives += column sums of dy (the bias gradient of the FC that produced y).
    planes: Planes that receive dy's bf16x3 operand planes INSTEAD of the fp32 values (the returned dy is then an unwritten token)."""
    L = _lib.lib()
    dev = y.device
    dy = torch.empty(N, d, dtype=torch.float32, device=dev)
    acc = dg_out is not None
    dg = dg_out if acc else torch.empty(d, dtype=torch.float32, device=dev)
    db = db_out if acc else torch.empty(d, dtype=torch.float32, device=dev)
    wsb = L.advmil_ln_relu_mean16_bwd_workspace_bytes(N, d)
    ws = _ws(wsb, dev)
    _lib.check(L.advmil_ln_relu_mean16_bwd(_p(demb), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), N, d, _p(None if planes is not None else dy),
                                           _p(dg), _p(db), 1 if acc else 0, _p(ycol_out), _p(None if planes is None else planes.hi),
                                           _p(None if planes is None else planes.lo), int(dup), _p(ws), wsb, _stream()), "ln_relu_mean16_bwd")
    return dy, dg, db


def adam_blocks(n):
    """Workgroups of the Adam launch over n elements = entries of its `abs_partial` output."""
    return int(_lib.lib().advmil_adam_blocks(int(n)))


def adam_step(p, grad, m, v, wd, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l1_coef=0.0, planes=None, tick=True,
              abs_partial=None, clear_grad=False):
    """In place over flat fp32 arenas; `step` is an int32 device tensor bumped by the kernel. `planes`: Planes arenas that receive
    the bf16x3 operand planes of the updated weights. abs_partial [adam_blocks(n)]: per-workgroup shares of sum |p| before the update;
    clear_grad: zero the gradient arena behind its last read."""
    if abs_partial is not None and (abs_partial.numel() < adam_blocks(p.numel()) or abs_partial.dtype != torch.float32):
        raise ValueError("adam_step: abs_partial needs adam_blocks(n) fp32 entries")
    _lib.check(_lib.lib().advmil_adam_step(_p(p), _p(grad), _p(m), _p(v), _p(wd), p.numel(), lr, beta1, beta2, eps,
                                           grad_scale, l1_coef, _p(step), _p(None if planes is None else planes.hi),
                                           _p(None if planes is None else planes.lo), 1 if tick else 0, _p(abs_partial),
                                           1 if clear_grad else 0, _stream()),
               "adam_step")


def step_seed_tick(step, seed, inc=1, step2=None):
    """step[0] += 1, step2[0] += 1 and seed[0] += inc in one launch (any may be None)."""
    _lib.check(_lib.lib().advmil_step_seed_tick(_p(step), _p(step2), _p(seed), int(inc), _stream()), "step_seed_tick")


def abs_sum(p):
    L = _lib.lib()
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    wsb = L.advmil_abs_sum_workspace_bytes(p.numel())
    ws = _ws(wsb, p.device)
    _lib.check(L.advmil_abs_sum(_p(p), p.numel(), _p(out), _p(ws), wsb, _stream()), "abs_sum")
    return out


# ---------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------
def _arena_grad(p):
    """The parameter's slot in its optimizer's flat gradient arena (set by advmil_amd.optim.FlatAdam), or None.
    When present, backward kernels ADD the parameter gradient straight into it and hand autograd `None`, which removes
    one accumulate launch per parameter per bag."""
    g = getattr(p, "_arena_grad", None) if p is not None else None
    if g is not None and FUSED_WGRAD and p.requires_grad:
        o = getattr(p, "_arena_owner", None)
        if o is not None and o._grad_clean and not torch.cuda.is_current_stream_capturing():
            o._grad_clean = False            # a kernel is about to add into the arena: it is no longer known to be all zero (optim.FlatAdam)
        return g
    return None


# storages of the optimizers' flat gradient arenas (optim.FlatAdam registers them): the only destinations a DEFERRED accumulate may have
ARENA_STORAGES = set()


def _is_arena(t):
    return t.untyped_storage().data_ptr() in ARENA_STORAGES


def _adjacent(a, b, itemsize=4):
    return (a is not None and b is not None and a.is_contiguous() and b.is_contiguous()
            and b.data_ptr() == a.data_ptr() + a.numel() * itemsize
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())     # same arena, not allocator luck


def _stack2(a, b, rows, cols):
    """[2*rows, cols] view over two adjacent arena tensors (attention_a|attention_b weights or biases), else a copy."""
    if _adjacent(a, b):
        shape = (2 * rows, cols) if cols else (2 * rows,)
        stride = (cols, 1) if cols else (1,)
        return a.detach().as_strided(shape, stride, a.storage_offset()), True
    return torch.cat([a.detach(), b.detach()], dim=0).contiguous(), False


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on a small dense tensor as one launch each way (flat index i on stream `sid`)."""

    @staticmethod
    def forward(ctx, x, p, seed, sid, rr=None):
        x = x.contiguous()
        y = torch.empty_like(x)
        w = x.shape[-1] if rr is not None else 0
        _lib.check(_lib.lib().advmil_dropout_apply(_p(x), _p(y), x.numel(), p, _p(seed), sid, _p(rr), w, _stream()), "dropout_apply")
        ctx.cfg = (p, seed, sid, rr, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, sid, rr, w = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().advmil_dropout_apply(_p(dy), _p(dx), dy.numel(), p, _p(seed), sid, _p(rr), w, _stream()), "dropout_apply")
        return dx, None, None, None, None


def dropout(x, p, rng, tag=""):
    """Train-mode dropout of x (any shape) drawn from the counter RNG at a fresh call site."""
    if p <= 0.0 or x.numel() == 0:
        return x
    _chk(x, "x")
    sid = rng.site(tag, tuple(x.shape), p)
    return DropoutFn.apply(x, float(p), rng.seed, sid, rng.row_map(x.numel() // x.shape[-1], tag))


class LinearActFn(torch.autograd.Function):
    """y = dropout(act(x W^T + b)); x[M,K], W[N,K]. Dropout index = m*N + n on stream `sid`."""
    last_planes = None       # planes of the y just produced (side channel to linear_act: Function outputs are re-wrapped)
    last_wants_dy_planes = False
    last_act_fusable = None
    last_maskbits = None
    gate_request = None      # (p_gate, tag_a, tag_b, rng) set by linear_act: the consumer is a gated attention pool in train mode
    last_gate = None         # (stream a, stream b, bits_a, bits_b, p_gate, row map): drawn with this layer's own dropout (dropout_planes)

    @staticmethod
    def forward(ctx, x, W, b, act, p, seed, sid, y0=None, rr=None, xpl=None, wpl=None, emit=False):
        if not is_bf16_slab(x):               # (a bf16 slab is read through its plane only: linear_act made sure of that)
            _chk(x, "x")
        _chk(W, "weight")
        x = x.contiguous()
        W2 = W.detach().reshape(W.shape[0], -1)
        M, K = x.shape
        N = W2.shape[0]
        cpl = None
        ctx.small = False
        # a memoized / prefilled output that exists as operand planes only (H_PLANES_ONLY): y0 is an unwritten token carrying them
        y0pl = getattr(y0, "_advmil_planes", None) if (y0 is not None and getattr(y0, "_advmil_planes_only", False)) else None
        if (SMALL_LINEAR and y0 is None and M <= min(32, SMALL_LINEAR_ROWS) and K % 4 == 0 and N <= 1024 and xpl is None and not emit and x.dtype == torch.float32
                and (b is None or b.dtype == torch.float32)):
            # a [B, d] head / tail layer: one fp32-FMA launch (csrc/optim.hip small_linear_*), its backward one or two
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            use = seed is not None and p > 0.0
            _lib.check(_lib.lib().advmil_small_linear_fwd(_p(x), K, _p(W2), _p(b), M, N, K, act, float(p) if use else 0.0,   # (x is contiguous: pitch K)
                                                          _p(seed if use else None), sid, _p(rr if use else None), _p(y), _stream()),
                       f"small_linear_fwd[{M}x{N}x{K}]")
            ctx.small = True
        elif y0 is None:
            if emit and get_gemm_mode() == "bf16x3":
                cpl = Planes.alloc((M, N), x.device)
            if emit == "only" and cpl is not None:
                # the consumer reads y as operand planes and nothing else does (the attention kernels behind the ESAT in-projection): the
                # epilogue writes the planes INSTEAD of the fp32 values; y is an unwritten token that carries shape and autograd identity
                gemm(x, W2, True, True, M, N, K, bias=b, act0=act, drop_p=p, seed=seed, stream_id=sid, rng_row=rr,
                     a_planes=xpl, b_planes=wpl, c_planes=cpl, c_planes_only=True)
                cpl.fp32_stale = True
                y = _token(M, N, x.device)
            else:
                y = gemm(x, W2, True, True, M, N, K, bias=b, act0=act, drop_p=p, seed=seed, stream_id=sid, rng_row=rr,
                         a_planes=xpl, b_planes=wpl, c_planes=cpl, splits=1 if cpl is not None else None)
        elif p > 0.0 and y0pl is not None:
            # ... memoized as planes: the draw maps planes to planes (+ the keep-and-positive bits), 8 bytes per element instead of 12
            greq, LinearActFn.gate_request = LinearActFn.gate_request, None
            gate = None
            if greq is not None and N % 32 == 0:
                pg, tag_a, tag_b, grng = greq
                if grng.row_map(M, tag_a) is rr and grng.row_map(M, tag_b) is rr:      # one row map for the three draws of the launch
                    # (the scorer's two sites are the next two draws anyway: the numbering is what gated_attn_pool would have produced)
                    gate = (pg, grng.site(tag_a, (M, N), pg), grng.site(tag_b, (M, N), pg))
            if gate is not None:
                cpl, mbits, gb = dropout_planes(y0pl, M, N, p, seed, sid, rr, gate=gate)
                LinearActFn.last_gate = (gate[1], gate[2], gb[0], gb[1], gate[0], rr)
            else:
                cpl, mbits = dropout_planes(y0pl, M, N, p, seed, sid, rr)
            y = _token(M, N, x.device)
            LinearActFn.last_maskbits = mbits if act == ACT_RELU else None
        elif p > 0.0:       # memoized act(x W^T + b) of the eval forward: only this forward's dropout draw is new
            if emit and MEMO_PLANES and get_gemm_mode() == "bf16x3":
                cpl = Planes.alloc((M, N), x.device)
            # (+ the ReLU-and-kept mask as one bit per element: the backward of this layer then never reads y back -- ACT_BWD_IN_DH)
            mbits = (torch.empty(M, N // 32, dtype=torch.int32, device=x.device)
                     if (ACT_BWD_IN_DH and act == ACT_RELU and N % 32 == 0 and M >= 4096 and get_gemm_mode() == "bf16x3"
                         and (xpl is not None or planes_of(x) is not None) and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]) else None)
            y, _ = act_dropout_bwd(y0, y0, ACT_NONE, M, N, p, seed, sid, want_bias=False, rng_row=rr, planes=cpl, bits=mbits)
            LinearActFn.last_maskbits = mbits
        else:
            y = y0
            cpl = y0pl
        LinearActFn.last_planes = cpl
        ctx.ypl = cpl if (cpl is not None and cpl.fp32_stale) else None      # y is a token: a backward that needs its values rebuilds them
        ctx.save_for_backward(x, W2, y)
        ctx.cfg = (act, p, seed, sid, M, N, K, W.shape, b is not None, rr)
        ctx.gW, ctx.gb = _arena_grad(W), _arena_grad(b)
        ctx.xpl = xpl if xpl is not None else (planes_of(x) if M >= 4096 else None)     # the weight gradient's big operand, pre-split
        # a plain slab layer (no activation, no dropout, constant bias, no input gradient) whose weight gradient takes both operands
        # pre-split: the backward of the LayerNorm behind it may hand dy over as operand planes only (ops.ln_relu_mean16 / DY_PLANES)
        ctx.wants_dy_planes = bool(
            not ctx.small and act == ACT_NONE and p <= 0.0 and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
            and not (b is not None and ctx.needs_input_grad[2]) and DG_PLANES_ONLY and DW_PLANES and get_gemm_mode() == "bf16x3"
            and ctx.xpl is not None and M >= 4096 and N % 8 == 0 and pre_a_tile_ok(gemm_plan(N, K, M, False, False)[0], False, False, True))
        LinearActFn.last_wants_dy_planes = ctx.wants_dy_planes
        # a ReLU (+ dropout) slab layer without an input gradient whose dpre feeds the plane-fed weight gradient alone: the consumer of y
        # (the gated-attention pool) may run this layer's activation / dropout backward in the epilogue of its own dh contraction
        # (ops.ACT_BWD_FUSED) and hand dpre over as operand planes with the bias gradient already merged
        ctx.act_fusable = bool(
            ACT_BWD_IN_DH and not ctx.small and act == ACT_RELU and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
            and DG_PLANES_ONLY and DW_PLANES and get_gemm_mode() == "bf16x3" and ctx.xpl is not None and M >= 4096 and N % 8 == 0
            and (not (b is not None and ctx.needs_input_grad[2]) or ctx.gb is not None)
            and pre_a_tile_ok(gemm_plan(N, K, M, False, False)[0], False, False, True))
        LinearActFn.last_act_fusable = (float(p), ctx.gb if (b is not None and ctx.needs_input_grad[2]) else None) if ctx.act_fusable else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2, y = ctx.saved_tensors
        act, p, seed, sid, M, N, K, wshape, has_b, rr = ctx.cfg
        dy = dy.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = has_b and ctx.needs_input_grad[2]
        db = None
        if ctx.small:
            L = _lib.lib()
            dev = x.device
            acc_w, acc_b = need_w and ctx.gW is not None, need_b and ctx.gb is not None
            dW = (ctx.gW.view(N, K) if acc_w else torch.empty(N, K, dtype=torch.float32, device=dev)) if need_w else None
            db = (ctx.gb if acc_b else torch.empty(N, dtype=torch.float32, device=dev)) if need_b else None
            dx = torch.empty(M, K, dtype=torch.float32, device=dev) if need_x else None
            wsb = L.advmil_small_linear_bwd_workspace_bytes(M, N) if need_x else 0
            ws = _ws(wsb, dev) if wsb else None
            use = seed is not None and p > 0.0
            _lib.check(L.advmil_small_linear_bwd(_p(dy), _p(y), _p(x), K, _p(W2), M, N, K, act, float(p) if use else 0.0,
                                                 _p(seed if use else None), sid, _p(rr if use else None), _p(dW), 1 if acc_w else 0, _p(db),
                                                 1 if acc_b else 0, _p(dx), K, _p(ws), wsb, _stream()), f"small_linear_bwd[{M}x{N}x{K}]")
            return (dx, None if (dW is None or acc_w) else dW.reshape(wshape), None if (db is None or acc_b) else db,
                    None, None, None, None, None, None, None, None, None)
        if act == ACT_NONE and p <= 0.0:
            ent = DY_PLANES.pop(dy) if ctx.wants_dy_planes else None
            if ent is not None and ent[1] == (M, N) and need_w:
                # dy arrived as operand planes only (written by the LayerNorm backward): dW = dy^T X with both operands pre-split
                dpl, xpl0 = ent[0], ctx.xpl
                if ctx.gW is not None:
                    gemm(None, x, False, False, N, K, M, out=ctx.gW.view(N, K), ldc=K, accumulate=True, a_planes=dpl, b_planes=xpl0)
                    return None, None, None, None, None, None, None, None, None, None, None, None
                dW = gemm(None, x, False, False, N, K, M, a_planes=dpl, b_planes=xpl0).reshape(wshape)
                return None, dW, None, None, None, None, None, None, None, None, None, None
            if ctx.wants_dy_planes and DY_PLANES.pending((M, N)):
                raise RuntimeError("advmil_amd: a gradient written as operand planes only did not reach the layer that asked for it "
                                   "(autograd re-wrapped the token tensor?)")
            dpre = dy
            if need_b:
                db = colsum(dy, M, N, out=ctx.gb)
        else:
            ent = DY_PLANES.pop(dy) if ctx.act_fusable else None
            if ent is not None and ent[1] == (M, N) and ent[2] == "dpre":
                # the consumer of y already applied this layer's activation / dropout backward in its dh contraction's epilogue: `dy` is a
                # token, dpre arrives as operand planes, the bias gradient is merged: only the weight gradient is left
                dpl, xpl0 = ent[0], ctx.xpl
                if ctx.gW is not None:
                    gemm(None, x, False, False, N, K, M, out=ctx.gW.view(N, K), ldc=K, accumulate=True, a_planes=dpl, b_planes=xpl0)
                    return None, None, None, None, None, None, None, None, None, None, None, None
                dW = gemm(None, x, False, False, N, K, M, a_planes=dpl, b_planes=xpl0).reshape(wshape)
                return None, dW, None, None, None, None, None, None, None, None, None, None
            # slab layer whose input needs no gradient (the first layer): dpre is consumed by the weight-gradient contraction alone,
            # which splits it into hi + lo anyway -> written as planes only, and the contraction takes both operands pre-split
            xpl0 = ctx.xpl if ((DW_PLANES or (ctx.xpl is not None and ctx.xpl.fp32_stale)) and get_gemm_mode() == "bf16x3") else None
            only = (DG_PLANES_ONLY and need_w and not need_x and xpl0 is not None and M >= 4096 and N % 8 == 0
                    and pre_a_tile_ok(gemm_plan(N, K, M, False, False)[0], False, False, True))
            dpl = Planes.alloc((M, N), dy.device) if only else None
            if getattr(ctx, "ypl", None) is not None:
                y = planes_f32(ctx.ypl)       # (the consumer did not take the activation backward into its own epilogue: rare path)
            dpre, db = act_dropout_bwd(dy, y, act, M, N, p, seed, sid, want_bias=need_b, db_out=ctx.gb if need_b else None, rng_row=rr,
                                       planes=dpl, planes_only=only)
            if only:
                if ctx.gW is not None:
                    gemm(None, x, False, False, N, K, M, out=ctx.gW.view(N, K), ldc=K, accumulate=True, a_planes=dpl, b_planes=xpl0)
                    return None, None, (None if ctx.gb is not None else db), None, None, None, None, None, None, None, None, None
                dW = gemm(None, x, False, False, N, K, M, a_planes=dpl, b_planes=xpl0).reshape(wshape)
                return None, dW, (None if ctx.gb is not None else db), None, None, None, None, None, None, None, None, None
        dW = None
        if need_w:                                           # dpre^T x
            xpl = ctx.xpl if ((DW_PLANES or (ctx.xpl is not None and ctx.xpl.fp32_stale)) and get_gemm_mode() == "bf16x3") else None
            if xpl is None and is_bf16_slab(x):
                x = as_f32(x)
            if ctx.gW is not None:
                gemm(dpre, x, False, False, N, K, M, out=ctx.gW.view(N, K), ldc=K, accumulate=True, b_planes=xpl)
            else:
                dW = gemm(dpre, x, False, False, N, K, M, b_planes=xpl).reshape(wshape)
        dx = gemm(dpre, W2, True, False, M, K, N) if need_x else None                   # dpre W
        return dx, dW, (None if ctx.gb is not None else db), None, None, None, None, None, None, None, None, None


class ForwardMemo:
    """Reuse of row-sized pre-dropout layer outputs between two forwards over the SAME rows with the SAME weights.

    The reference runs the generator twice per optimizer step on every bag -- netG.eval() under no_grad for the discriminator
    update (model_handler.py:398-400), netG.train() for the generator update (model_handler.py:420-425) -- and the generator's
    weights do not change in between. act(x W^T + b) of a row-sized layer is therefore identical in both; only the dropout draw
    differs. mode 'record' (the no-grad eval forward) keeps each such output; mode 'replay' (the train forward) takes it back,
    applies its own dropout mask elementwise and rebuilds the same autograd node, instead of repeating the contraction
    (131072x384x1024: ~0.45 ms saved per step). `token` must identify the weights' version; entries from another token are dropped."""

    def __init__(self):
        self.mode, self.token, self.rows_ptr, self.store = None, None, None, {}
        # round 6 -- the chain continues behind the first layer for as long as nothing random happens: `derived` holds the tensors the
        # record pass produced from the slab by deterministic ops alone (address -> the tensor itself: the memo keeps them ALIVE, so the
        # allocator cannot hand the address to anything else before the replay has taken them back); a layer applied to one of them is
        # carried like a layer applied to the slab (ESAT: FC -> LayerNorm/ReLU/mean16 -> in-projection, all of it before the first
        # dropout of the transformer layer). `ln`: {(y address, gamma address + versions, dup): (emb, mean, rstd)} of ops.ln_relu_mean16.
        self.derived, self.ln = {}, {}

    def begin(self, mode, token, rows):
        """`rows`: the step slab. Only layers applied to it -- or to a tensor the record pass DERIVED from it without any random draw
        (`derived`) -- are carried: any other intermediate tensor's address says nothing about its contents (the allocator reuses
        addresses)."""
        if token != self.token:
            self.store.clear(); self.derived.clear(); self.ln.clear()
        self.mode, self.token, self.rows_ptr = mode, token, rows.data_ptr()

    def knows(self, x):
        return self.mode is not None and (x.data_ptr() == self.rows_ptr or (MEMO_CHAIN and x.data_ptr() in self.derived))

    def end(self, clear=False):
        self.mode = None
        if clear:
            self.store.clear(); self.derived.clear(); self.ln.clear()


MEMO = ForwardMemo()
MEMO_CHAIN = os.environ.get("ADVMIL_MEMO_CHAIN", "1") != "0"
# only slab-sized layers are worth carrying (ADVMIL_MEMO_MIN_ROWS=1000000000 turns the memo off, for A/B timing)
MEMO_MIN_ROWS = int(os.environ.get("ADVMIL_MEMO_MIN_ROWS", "4096"))


# Outputs computed ahead of the layer call that owns them: {(rows ptr, shape, weight ptr, weight version, act): (y, planes of y)}.
# `prefill_two_layers` fills it with BOTH first layers over a step slab from one launch (the generator's and the discriminator's:
# X is staged once instead of twice); `linear_act` takes an entry exactly like a memo replay. The handler clears it per step.
PREFILL = {}
TWO_LAYERS = os.environ.get("ADVMIL_TWO_LAYERS", "1") != "0"


def prefill_two_layers(X, layer1, layer2):
    """layer = (weight [N, K], bias | None, act name, emit planes of the output?). Launches both layers over the slab X when the
    shapes and the resident planes allow it and parks the results for the two `linear_act` calls to come. -> True if it did."""
    (W1, b1, a1, emit1), (W2, b2, a2, _) = layer1, layer2       # the PARAMETERS themselves (their planes hang on them)
    if not TWO_LAYERS or X.dim() != 2 or not X.is_contiguous():
        return False
    M, K = X.shape
    W1m, W2m = W1.reshape(W1.shape[0], -1), W2.reshape(W2.shape[0], -1)
    if W2m.shape[1] != K or W1m.shape[1] != K or not gemm_two_layers_ok(M, W1m.shape[0], W2m.shape[0], K):
        return False
    xpl, p1, p2 = planes_of(X), weight_planes(W1), weight_planes(W2)
    if xpl is None or p1 is None or p2 is None:
        return False
    emit = bool(emit1) and bool(gemm_plan_planes(M, 2 * W1m.shape[0], W1m.shape[0]))
    # layer 1 = the generator's ReLU layer in front of the gated-attention pool: every consumer of its output reads operand planes
    only = bool(emit and H_PLANES_ONLY and a1 == "relu" and W1m.shape[0] % 32 == 0)
    with torch.no_grad():
        y1, y2, cpl = gemm_two_layers(X, xpl, W1m.detach(), Planes(p1.hi.reshape(W1m.shape), p1.lo.reshape(W1m.shape)),
                                      None if b1 is None else b1.detach(), _ACT[a1],
                                      W2m.detach(), Planes(p2.hi.reshape(W2m.shape), p2.lo.reshape(W2m.shape)),
                                      None if b2 is None else b2.detach(), _ACT[a2], emit, y1_planes_only=only)
    if only:
        y1._advmil_planes, y1._advmil_planes_only = cpl, True
    PREFILL[(X.data_ptr(), tuple(X.shape), W1.data_ptr(), W1._version, _ACT[a1])] = (y1, cpl)
    PREFILL[(X.data_ptr(), tuple(X.shape), W2.data_ptr(), W2._version, _ACT[a2])] = (y2, None)
    return True


def linear_act(x, W, b, act="none", p=0.0, rng=None, tag="", emit_planes=False, gate_sites=None):
    """x[..., K] -> [..., N] through the HIP GEMM (any leading dims are flattened). In bf16x3 mode the operands' bf16 planes are
    used when they exist (slab registered by the handler / producer-emitted activation planes / arena weight planes), and
    `emit_planes` makes the epilogue also write the planes of y (attribute `_advmil_planes`) for the contraction that reads it."""
    lead = x.shape[:-1]
    x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])        # (a 2-D input keeps its object: operand planes are attributes)
    sid, seed, rr = 0, None, None
    if p > 0.0:
        rng = rng or default_rng(x.device)
        N = W.shape[0]
        sid, seed, rr = rng.site(tag, (x2.shape[0], N), p), rng.seed, rng.row_map(x2.shape[0], tag)
    # (the slab itself never needs a gradient; a derived tensor does in the replay -- the layer's backward only needs its INPUT, not how
    # the output was obtained)
    memo = MEMO if (MEMO.knows(x2) and x2.shape[0] >= MEMO_MIN_ROWS and (not x2.requires_grad or x2.data_ptr() != MEMO.rows_ptr)) else None
    # W._version: load_state_dict / any in-place torch write to the weight invalidates the entry (the fused Adam kernel writes
    # through raw pointers, which the memo's token -- the optimizer's update count -- covers)
    key = (x2.data_ptr(), tuple(x2.shape), W.data_ptr(), W._version, act) if memo is not None else None
    y0 = None
    pre_planes = None
    if PREFILL and p <= 0.0:
        pf = PREFILL.pop((x2.data_ptr(), tuple(x2.shape), W.data_ptr(), W._version, _ACT[act]), None)
        if pf is not None:
            y0, pre_planes = pf
    if y0 is None and memo is not None and memo.mode == "replay":
        y0 = memo.store.pop(key, None)
        if y0 is not None:
            y0.record_stream(torch.cuda.current_stream())      # recorded on one stream, replayed (and released) on another
    xpl = wpl = None
    if y0 is not None and p <= 0.0 and emit_planes:
        emit_planes = False                   # the memoized tensor is returned as is: its planes (if any) are already attached
    big = x2.shape[0] >= 4096 and get_gemm_mode() == "bf16x3"
    stale = bool(getattr(x2, "_advmil_fp32_stale", False))      # a slab staged as operand planes only: its fp32 rows were never written
    if stale and (planes_of(x2) is None or get_gemm_mode() != "bf16x3"):
        raise RuntimeError("advmil_amd: a step slab staged as operand planes only reached a layer without its planes")
    # (GENERIC_PLANES: a slab layer whose launch stays on the generic kernel still takes pre-split operands when both exist -- the
    # 64x64 .. 128x128 tiles are built for them: no conversion work in the staging path)
    if (y0 is None and big and (gemm_plan_planes(x2.shape[0], W.shape[0], x2.shape[1]) or (GENERIC_PLANES and planes_of(x2) is not None))) or stale:
        xpl, wpl = planes_of(x2), weight_planes(W)
        if wpl is not None:
            wpl = Planes(wpl.hi.reshape(W.shape[0], -1), wpl.lo.reshape(W.shape[0], -1))
    if is_bf16_slab(x2) and get_gemm_mode() != "bf16x3":
        x2 = as_f32(x2)                       # exact-fp32 arithmetic reads fp32 operands: the slab's (exact) fp32 image
    # emit y's planes only when the contraction that reads y (the gate branches: N' = 2N columns over K' = N) will take them
    emit = bool(emit_planes) and big and bool(gemm_plan_planes(x2.shape[0], 2 * W.shape[0], W.shape[0]))
    if emit_planes == "only":                 # planes INSTEAD of fp32 values (the caller guarantees a planes-only consumer): any slab-sized layer
        # (an activated layer only where nothing will differentiate through it: its backward would need the values)
        emit = "only" if (big and (act == "none" or not torch.is_grad_enabled()) and p <= 0.0 and y0 is None and W.shape[0] % 8 == 0
                          and ATTN_QKV_PLANES and (act == "none" or emit)) else (emit if act != "none" else False)
    # gate_sites = (p_gate, tag_a, tag_b): the output feeds a gated attention pool that will draw these two dropout sites next; when this
    # call turns out to be the planes-to-planes dropout replay, their keep bits are drawn in the same launch (FUSED_GATE_TRAIN)
    LinearActFn.gate_request = ((float(gate_sites[0]), gate_sites[1], gate_sites[2], rng) if (gate_sites is not None and FUSED_GATE_TRAIN
                                and p > 0.0 and gate_sites[0] > 0.0 and torch.is_grad_enabled()) else None)
    LinearActFn.last_gate = None
    y = LinearActFn.apply(x2, W, b, _ACT[act], float(p), seed, sid, y0, rr, xpl, wpl, emit)
    LinearActFn.gate_request = None
    cpl, LinearActFn.last_planes = LinearActFn.last_planes, None
    if pre_planes is not None and cpl is None:
        cpl = pre_planes                      # the two-layer launch already emitted the planes of this output
    if memo is not None and memo.mode == "record" and p <= 0.0 and not torch.is_grad_enabled():
        memo.store[key] = y
        memo.derived[y.data_ptr()] = y
    out = y if len(lead) == 1 else y.reshape(*lead, y.shape[-1])
    if cpl is not None:
        out._advmil_planes = cpl
        if emit == "only" or cpl.fp32_stale:
            out._advmil_planes_only = True    # the fp32 values of `out` were never written
    if LinearActFn.last_wants_dy_planes:
        out._advmil_wants_dy_planes = True         # the LayerNorm backward behind this layer may hand dy over as operand planes only
    LinearActFn.last_wants_dy_planes = False
    if LinearActFn.last_act_fusable is not None and out.requires_grad:
        # (dropout rate, bias-gradient slot | None, bit mask | None): see GatedAttnPoolFn.backward
        out._advmil_act_fusable = LinearActFn.last_act_fusable + (LinearActFn.last_maskbits,)
    LinearActFn.last_act_fusable = None
    LinearActFn.last_maskbits = None
    if LinearActFn.last_gate is not None:
        out._advmil_gate = LinearActFn.last_gate
        LinearActFn.last_gate = None
    return out


def linear_act_any(x, W, b, act="none", p=0.0, rng=None, tag=""):
    """linear_act for a [B, d]-sized layer of ANY width: the contraction engine wants every contiguous extent to be a multiple of 4
    floats, so a layer that is not (no shipped configuration has one; odd `pdh_dims` / `hid_dims` of the reference's builders do)
    runs on zero-padded copies of x / W / b (torch pad + slice: copies, no arithmetic) and its dropout is drawn on the unpadded
    [B, N] result, i.e. from the same (tag, shape) site as before. There is no eager-ATen linear anywhere in the product path."""
    K, N = W.shape[1], W.shape[0]
    if x.dim() != 2 or not x.is_cuda:
        raise RuntimeError("advmil_amd: linear layers run on 2-D HIP tensors only (no CPU / eager fallback)")
    if K % 4 == 0 and N % 4 == 0:
        return linear_act(x, W, b, act, p, rng, tag)
    pk, pn = (-K) % 4, (-N) % 4
    xp = torch.nn.functional.pad(x, (0, pk)) if pk else x
    Wp = torch.nn.functional.pad(W, (0, pk, 0, pn)) if (pk or pn) else W
    bp = None if b is None else (torch.nn.functional.pad(b, (0, pn)) if pn else b)
    y = linear_act(xp.contiguous(), Wp.contiguous(), bp, act)[:, :N]
    if p > 0.0:
        y = dropout(y.contiguous(), p, rng or default_rng(x.device), tag)
    return y


class GatedAttnPoolFn(torch.autograd.Function):
    """(pooled[B,D], A[N]) = per-bag softmax-pool of h[N,D] scored by the gated attention net; `seg` partitions the rows
    into the B bags of a step slab (None = one bag). Attn_Net_Gated + softmax + mm (model/backbone_utils.py:11-29,
    model/backbone.py:81-85) and GAPool (model/backbone_utils.py:47-56): the pooled tensor is the scored tensor in every use."""

    @staticmethod
    def forward(ctx, h, Wa, ba, Wb, bb, wc, bc, p, seed, sa, sb, seg, nograd=False, rr=None, hpl=None, act_fuse=None, gate=None):
        _chk(h, "h")
        h = h.contiguous()
        N, D = h.shape
        ctx.act_fuse = act_fuse
        ctx.pair32 = False
        wcv = wc.detach().reshape(-1)
        stale = hpl is not None and hpl.fp32_stale            # h is a token: every read below goes through its planes
        ppl = hpl if stale else None
        if FUSED_GATE_SCORE and p <= 0.0 and N >= 4096 and nograd:
            # no-grad pass (the generator's eval forward of the discriminator update, test_model): nothing needs the [N, 2D] gate
            # activations, so the contraction reduces the score in its epilogue from interleaved branch rows and never stores them
            Wi, bi, wipl = gate_interleave(Wa.detach(), ba.detach(), Wb.detach(), bb.detach(), D, planes=hpl is not None)   # one tiny launch
            s = gate_partial_sum(gemm(h, Wi, True, True, N, 2 * D, D, bias=bi, gate_wc=wcv, a_planes=hpl, b_planes=wipl), bc.detach())
            A, pooled = softmax_pool(s, h, N, D, seg, ppl)
            ctx.mark_non_differentiable(s)
            return pooled, A, s
        # fused-weight-gradient slots first: the training form below needs them (its dWab rows come out in pair-block order and only the
        # accumulating merge un-permutes them)
        gs = [_arena_grad(t) for t in (Wa, ba, Wb, bb, wc, bc)]
        arena_ok = all(g is not None for g in gs) and _adjacent(gs[0], gs[2]) and _adjacent(gs[1], gs[3])
        if (gate is not None and FUSED_GATE_TRAIN and p > 0.0 and hpl is not None and arena_ok and N % 256 == 0 and D % 64 == 0
                and get_gemm_mode() == "bf16x3" and gemm_plan_planes(N, 2 * D, D) and gemm_plan_tn_planes(2 * D, D, N)[1] > 1
                and gemm_plan_planes(N, D, 2 * D)):
            # TRAINING pass with the score in the contraction's epilogue: branches in pair blocks of 32 columns, keep bits from `gate`
            Wp, bp, wppl = gate_interleave(Wa.detach(), ba.detach(), Wb.detach(), bb.detach(), D, planes=True, pair32=True)
            ab, part = gemm(h, Wp, True, True, N, 2 * D, D, bias=bp, gate_wc=wcv, drop_p=p, a_planes=hpl, b_planes=wppl, gate_bits=gate)
            s = gate_partial_sum(part, bc.detach())
            A, pooled = softmax_pool(s, h, N, D, seg, ppl)
            ctx.save_for_backward(h, Wp, ab, A, wcv)
            ctx.hpl, ctx.wabpl, ctx.pair32 = hpl, wppl, True
            ctx.cfg = (p, seed, sa, sb, N, D, wc.shape, seg, rr)
            ctx.arena = (gs[0].as_strided((2 * D, D), (D, 1), gs[0].storage_offset()),
                         gs[1].as_strided((2 * D,), (1,), gs[1].storage_offset()), gs[4].view(-1), gs[5])
            ctx.mark_non_differentiable(s)
            ctx.set_materialize_grads(False)
            return pooled, A, s
        Wab, _ = _stack2(Wa, Wb, D, D)                       # [2D, D]: a view when the two live side by side in the arena
        bab, _ = _stack2(ba, bb, D, 0)
        wabpl = None
        if hpl is not None:
            pa_, pb_ = weight_planes(Wa), weight_planes(Wb)
            if pa_ is not None and pb_ is not None and _adjacent(pa_.hi, pb_.hi, 2) and _adjacent(pa_.lo, pb_.lo, 2):
                wabpl = Planes(pa_.hi.as_strided((2 * D, D), (D, 1), pa_.hi.storage_offset()),
                               pa_.lo.as_strided((2 * D, D), (D, 1), pa_.lo.storage_offset()))
            if wabpl is None and stale:
                wabpl = split_planes(Wab.detach().contiguous())
        ab = gemm(h, Wab, True, True, N, 2 * D, D, bias=bab, act0=ACT_TANH, act1=ACT_SIGMOID, act_split=D,
                  a_planes=hpl if wabpl is not None else None, b_planes=wabpl)
        s = gate_score(ab, wcv, bc, N, D, p, seed, sa, sb, rr)
        A, pooled = softmax_pool(s, h, N, D, seg, ppl)
        ctx.save_for_backward(h, Wab, ab, A, wcv)
        ctx.hpl = hpl                          # h's operand planes: B operand of dWab = dG^T h (with dG as planes: the plane-fed TN kernel)
        ctx.wabpl = wabpl                      # Wab's planes: transposed, the B operand of dh = dG Wab on the plane-fed NT kernel
        ctx.cfg = (p, seed, sa, sb, N, D, wc.shape, seg, rr)
        # fused weight-gradient accumulation needs every parameter's arena slot, with the a|b pairs adjacent
        ctx.arena = None
        if arena_ok:
            ctx.arena = (gs[0].as_strided((2 * D, D), (D, 1), gs[0].storage_offset()),
                         gs[1].as_strided((2 * D,), (1,), gs[1].storage_offset()), gs[4].view(-1), gs[5])
        ctx.mark_non_differentiable(s)
        ctx.set_materialize_grads(False)
        return pooled, A, s

    @staticmethod
    def backward(ctx, dpooled, dA, _ds_unused):
        h, Wab, ab, A, wcv = ctx.saved_tensors               # (pair32: Wab is the pair-block stack of the two branches, and so are ab / dG)
        p, seed, sa, sb, N, D, wcshape, seg, rr = ctx.cfg
        pair32 = bool(getattr(ctx, "pair32", False))
        nseg = 1 if seg is None else seg.nseg
        dpooled = (torch.zeros(nseg, D, dtype=torch.float32, device=h.device) if dpooled is None
                   else dpooled.contiguous().reshape(nseg, D))
        dA_ = None if dA is None else dA.contiguous()
        stale = ctx.hpl is not None and ctx.hpl.fp32_stale    # h is a token (planes only)
        ds = softmax_pool_bwd(dpooled, dA_, A, h, N, D, seg, ctx.hpl if stale else None)
        need_h = ctx.needs_input_grad[0]
        # bf16x3: dh = dG Wab runs as an NT contraction of dG's planes (emitted by gate_bwd) with the planes of Wab^T (a 2D x D
        # transpose + split: two tiny launches) through the plane-fed kernel, when the shape qualifies
        gpl = None
        # (round 6: with the first layer's activation backward in its epilogue -- ctx.act_fuse -- the launch no longer stays on the generic
        # kernel: the plane-fed kernel's full epilogue carries the rank-1 + bit-mask + column-sum form too)
        dh_tile = gemm_plan_planes(N, D, 2 * D) if (need_h and USE_PLANES and get_gemm_mode() == "bf16x3") else 0
        dh_nt = bool(dh_tile and DH_NT_FUSED and ctx.act_fuse is not None and ctx.act_fuse[2] is not None
                     and getattr(ctx, "wabpl", None) is not None and int(_lib.lib().advmil_gemm_f32_colsum_rows(dh_tile, N, D)) > 0)
        # bf16x3, slab-sized: dG is consumed by exactly two contractions (dh = dG Wab, dWab = dG^T h) that would split it into hi + lo
        # anyway -> the gate backward writes the planes INSTEAD of the fp32 values (same bytes) and both take their A operand pre-split
        only = (DG_PLANES_ONLY and USE_PLANES and get_gemm_mode() == "bf16x3" and N >= 4096 and (2 * D) % 8 == 0
                and pre_a_tile_ok(gemm_plan(2 * D, D, N, False, False)[0], False, False)
                and (not need_h or pre_a_tile_ok(gemm_plan(N, D, 2 * D, True, False)[0], True, False)))
        if only and gpl is None:
            gpl = Planes.alloc((N, 2 * D), h.device)
        if ctx.arena is not None:
            gWab, gbab, gwc, gbc = ctx.arena
            dG, _, _, _ = gate_bwd(ab, ds, wcv, N, D, p, seed, sa, sb, dwc=gwc, dbc=gbc, dbias=gbab, rng_row=rr, planes=gpl, planes_only=only,
                                   pair32=pair32)
        else:
            dG, dwc, dbc, dbias = gate_bwd(ab, ds, wcv, N, D, p, seed, sa, sb, rng_row=rr, planes=gpl, planes_only=only)
        dh = None
        fuse = ctx.act_fuse if (need_h and only and ctx.act_fuse is not None) else None
        if fuse is not None and stale and fuse[2] is None:
            fuse = None                          # (no bit mask and no rows of h to read the mask back from: the layer runs its own backward)
        wtpl = None
        if fuse is not None and dh_nt:
            # the bit-mask form only (a mask read back from h would be a second prefetched operand: the generic kernel's loop)
            nrow = int(_lib.lib().advmil_gemm_f32_colsum_rows(dh_tile, N, D))
            tile = dh_tile
            wtpl = planes_transposed(ctx.wabpl)
        elif fuse is not None:
            tile = gemm_plan(N, D, 2 * D, True, False)[0]
            nrow = int(_lib.lib().advmil_gemm_f32_colsum_rows(tile, N, D)) if pre_a_tile_ok(tile, True, False) else 0
            if nrow <= 0:
                fuse = None
        if fuse is not None:
            # dpre = (dG Wab + A dpooled) * (h > 0 ? 1 / (1 - p) : 0) -- h is the first layer's stored (post-dropout) output, so the mask IS
            # its ReLU and dropout backward -- written as operand planes only, its column sums (that layer's bias gradient) as per-wave
            # partial rows merged into the arena slot: the row pass act_dropout_bwd over dh and h (0.11 ms at the 16-bag slab) is gone
            p1, gb1, mbits = fuse
            dpl = Planes.alloc((N, D), h.device)
            cws = _ws(nrow * D * 4, h.device) if gb1 is not None else None
            # the mask: the layer's bit mask when its forward left one (1/32 of the bytes, parked in LDS by the epilogue), else h itself
            if wtpl is not None:             # NT over planes: A = dG [N, 2D], B = Wab^T [D, 2D], both k-contiguous
                gemm(None, None, True, True, N, D, 2 * D, rowv=A, colv=dpooled, rowseg=None if seg is None else seg.rowseg, a_planes=gpl,
                     b_planes=wtpl, maskbits=mbits, mask_scale=1.0 / (1.0 - p1) if p1 > 0.0 else 1.0, c_planes=dpl, c_planes_only=True,
                     colsum=cws, tile=tile)
            else:
                gemm(None, Wab, True, False, N, D, 2 * D, rowv=A, colv=dpooled, rowseg=None if seg is None else seg.rowseg, a_planes=gpl,
                     maskref=h if mbits is None else None, maskbits=mbits, mask_scale=1.0 / (1.0 - p1) if p1 > 0.0 else 1.0, c_planes=dpl,
                     c_planes_only=True, colsum=cws, tile=tile)
            if gb1 is not None:
                _lib.check(_lib.lib().advmil_merge_partials(_p(cws), nrow, D, D, _p(gb1), 1, _stream()), "merge_partials")
            dh = torch.empty(N, D, dtype=torch.float32, device=h.device)      # token: never written, never read
            DY_PLANES.put(dh, dpl, (N, D), "dpre")
        elif need_h:
            # dG [N,2D] . Wab [2D,D]  +  A[n] * dpooled[bag(n), d]   (pooling's direct path, rank-1 per bag)
            if only:
                dh = gemm(None, Wab, True, False, N, D, 2 * D, rowv=A, colv=dpooled, rowseg=None if seg is None else seg.rowseg, a_planes=gpl)
            else:
                dh = gemm(dG, Wab, True, False, N, D, 2 * D, rowv=A, colv=dpooled, rowseg=None if seg is None else seg.rowseg)
        nones = (None,) * 10
        apl = gpl if only else None
        if ctx.arena is not None:
            bpl = ctx.hpl if (stale or (apl is not None and gemm_plan_tn_planes(2 * D, D, N)[0])) else None
            if pair32 and (apl is None or bpl is None):
                raise RuntimeError("advmil_amd: the pair-block gate backward needs dG and h as operand planes")
            gemm(dG, h, False, False, 2 * D, D, N, out=gWab, ldc=D, accumulate=True, a_planes=apl, b_planes=bpl, c_rows_pair32=pair32)   # dG^T h
            return (dh, None, None, None, None, None, None) + nones
        bpl = ctx.hpl if (stale or (apl is not None and gemm_plan_tn_planes(2 * D, D, N)[0])) else None
        dWab = gemm(dG, h, False, False, 2 * D, D, N, a_planes=apl, b_planes=bpl)
        return (dh, dWab[:D], dbias[:D], dWab[D:], dbias[D:], dwc.reshape(wcshape), dbc) + nones


def gated_attn_pool(h, Wa, ba, Wb, bb, wc, bc, p=0.0, rng=None, tag="", seg=None):
    """Returns (pooled [nseg, D] -- [D] when seg is None --, A[N], raw scores[N])."""
    sa = sb = 0
    seed = rr = None
    gate = None
    if p > 0.0:
        rng = rng or default_rng(h.device)
        pre = getattr(h, "_advmil_gate", None)
        if pre is not None and pre[4] == float(p):
            # the two sites (and their keep bits) were drawn with the producing layer's own dropout: ops.linear_act(gate_sites=...)
            sa, sb, gate = pre[0], pre[1], (pre[2], pre[3])
            seed, rr = rng.seed, pre[5]
        else:
            sa = rng.site(tag + "att_a", tuple(h.shape), p)
            sb = rng.site(tag + "att_b", tuple(h.shape), p)
            seed, rr = rng.seed, rng.row_map(h.shape[0], tag + "att_a")
    # grad mode is always off INSIDE Function.forward, so "nothing here will be differentiated" is decided out here
    nograd = not torch.is_grad_enabled() or not any(t.requires_grad for t in (h, Wa, ba, Wb, bb, wc, bc))
    hpl = planes_of(h) if (h.shape[0] >= 4096 and h.is_contiguous() and get_gemm_mode() == "bf16x3"
                           and gemm_plan_planes(h.shape[0], 2 * h.shape[1], h.shape[1])) else None
    if getattr(h, "_advmil_planes_only", False):
        hpl = planes_of(h)                    # h is an unwritten token: its planes are all there is
        if hpl is None or not hpl.fp32_stale:
            raise RuntimeError("advmil_amd: a planes-only activation reached the gated-attention pool without its planes")
    pooled, A, s = GatedAttnPoolFn.apply(h, Wa, ba, Wb, bb, wc, bc, float(p), seed, sa, sb, seg, nograd, rr, hpl,
                                         getattr(h, "_advmil_act_fusable", None), gate)
    return (pooled[0] if seg is None else pooled), A, s


class LNReLUMean16Fn(torch.autograd.Function):
    """emb[N/16,d] = mean16(relu(LayerNorm(y))) -- tail of AVGPoolPatchEmbedding
    (model/backbone_utils.py:161-167)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps, ycol=None, dy_planes=False, dup=1, memo=None):
        _chk(y, "y")
        y = y.contiguous()
        N, d = y.shape
        ctx.dup = dup
        if memo is not None:                  # the record pass's result on the same y and parameters (ForwardMemo.ln): nothing to launch
            emb, mean, rstd = memo
            epl = None
        else:
            epl = None
            emb, mean, rstd = ln_relu_mean16_fwd(y, gamma, beta, N, d, eps, planes=epl, dup=dup)
        LNReLUMean16Fn.last_stats = (mean, rstd)
        LNReLUMean16Fn.last_planes = epl
        ctx.save_for_backward(y, gamma.detach(), beta.detach(), mean, rstd)
        gg, gb = _arena_grad(gamma), _arena_grad(beta)
        ctx.arena = (gg, gb) if (gg is not None and gb is not None) else None
        ctx.ycol = ycol
        ctx.dy_planes = bool(dy_planes)
        return emb

    @staticmethod
    def backward(ctx, demb):
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        N, d = y.shape
        # the FC that produced y takes dy as the A operand of its weight gradient dy^T X and nothing else reads dy (its bias gradient is
        # `ycol`): dy is then written as operand planes ONLY and handed over through DY_PLANES, keyed by the token tensor's address
        pl = Planes.alloc((N, d), y.device) if ctx.dy_planes else None
        if ctx.arena is not None:
            dy, dg, db = ln_relu_mean16_bwd(demb.contiguous(), y, gamma, beta, mean, rstd, N, d, ctx.arena[0], ctx.arena[1], ctx.ycol, planes=pl,
                                            dup=ctx.dup)
            dg = db = None
        else:
            dy, dg, db = ln_relu_mean16_bwd(demb.contiguous(), y, gamma, beta, mean, rstd, N, d, ycol_out=ctx.ycol, planes=pl, dup=ctx.dup)
        if pl is not None:
            DY_PLANES.put(dy, pl, (N, d))
        return dy, dg, db, None, None, None, None, None


# operand planes of a gradient that was written as planes only. The consumer (LinearActFn.backward of the FC that produced the normalised
# tensor) pops its entry; a token nobody claims is a bug and raises there.
class PlaneHandover:
    """Gradients that exist as operand planes only, on their way from the backward that produced them to the ONE backward that consumes
    them (autograd re-wraps gradient tensors, so attributes do not survive the trip; the fp32 `token` it carries is never written).
    Per THREAD (autograd runs a handler's backward on the calling thread: model_handler sets set_multithreading_enabled(False)), keyed
    by (device, address of the token) and holding the token itself -- while an entry lives the allocator cannot hand that address to
    another tensor, so a stale entry can never meet a reused address. A token nobody claims raises in the consumer (`pending`) and the
    handlers clear the table behind every backward. ADVMIL_POISON_TOKENS=1 (the GPU test suite sets it) fills every token with NaN, so
    that anything that READS one -- a hook, retain_grad, a second consumer -- shows up instead of computing on garbage."""

    def __init__(self):
        import threading
        self._tl = threading.local()

    def _d(self):
        d = getattr(self._tl, "d", None)
        if d is None:
            d = self._tl.d = {}
        return d

    def put(self, token, planes, shape, kind=None):
        if POISON_TOKENS:
            token.fill_(float("nan"))
        self._d()[(token.device.index, token.data_ptr())] = (planes, tuple(shape), kind, token)

    def pop(self, grad):
        """(planes, shape, kind) handed over under the tensor `grad` arrived as, or None."""
        ent = self._d().pop((grad.device.index, grad.data_ptr()), None)
        return None if ent is None else ent[:3]

    def pending(self, shape):
        return any(v[1] == tuple(shape) for v in self._d().values())

    def __len__(self):
        return len(self._d())

    def __bool__(self):
        return bool(self._d())

    def clear(self):
        self._d().clear()


POISON_TOKENS = os.environ.get("ADVMIL_POISON_TOKENS", "0") == "1"
DY_PLANES = PlaneHandover()
LN_DY_PLANES = os.environ.get("ADVMIL_LN_DY_PLANES", "1") != "0"


def ln_relu_mean16_dup_ok(d):
    """Can the region embedding leave its kernel duplicated ([emb; emb], dup = 2)? The 16-byte kernels only."""
    return d % 128 == 0 and os.environ.get("ADVMIL_LN4", "1") != "0" and os.environ.get("ADVMIL_LN_DUP", "1") != "0"


def ln_relu_mean16(y, gamma, beta, eps=1e-5, ycol_grad=None, dup=1):
    """`ycol_grad`: optional [d] gradient accumulator (an arena slot) that receives += column sums of dy in the backward -- the bias
    gradient of the FC that produced y, for callers that hand that FC a detached bias. When that FC marked its output
    (`_advmil_wants_dy_planes`, set by linear_act: slab-sized layer, no input gradient, no bias of its own, planes of X resident), dy is
    produced as operand planes only."""
    want = bool(LN_DY_PLANES and ycol_grad is not None and getattr(y, "_advmil_wants_dy_planes", False) and y.is_contiguous())
    # forward memo (ForwardMemo.derived / .ln): y came out of the record pass's chain -> LayerNorm/ReLU/mean16 of it is the same in the replay
    mk = hit = None
    if MEMO_CHAIN and MEMO.mode is not None and y.is_contiguous() and MEMO.knows(y):
        mk = (y.data_ptr(), tuple(y.shape), gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, float(eps), int(dup))
        if MEMO.mode == "replay":
            hit = MEMO.ln.pop(mk, None)
    emb = LNReLUMean16Fn.apply(y, gamma, beta, eps, ycol_grad, want, int(dup), hit)
    epl, LNReLUMean16Fn.last_planes = getattr(LNReLUMean16Fn, "last_planes", None), None
    stats, LNReLUMean16Fn.last_stats = getattr(LNReLUMean16Fn, "last_stats", None), None
    if mk is not None and MEMO.mode == "record" and not torch.is_grad_enabled() and stats is not None:
        MEMO.ln[mk] = (emb, stats[0], stats[1])
        MEMO.derived[emb.data_ptr()] = emb
    if epl is not None:
        emb._advmil_planes = epl
    return emb


class GateScoreFn(torch.autograd.Function):
    """Raw gated-attention scores s[N] only (the reference's Attn_Net_Gated.forward contract,
    model/backbone_utils.py:24-29); the fused pool above is what the backbones call."""

    @staticmethod
    def forward(ctx, h, Wa, ba, Wb, bb, wc, bc, p, seed, sa, sb):
        _chk(h, "h")
        h = h.contiguous()
        N, D = h.shape
        Wab, _ = _stack2(Wa, Wb, D, D)
        bab, _ = _stack2(ba, bb, D, 0)
        ab = gemm(h, Wab, True, True, N, 2 * D, D, bias=bab, act0=ACT_TANH, act1=ACT_SIGMOID, act_split=D)
        wcv = wc.detach().reshape(-1).contiguous()
        s = gate_score(ab, wcv, bc, N, D, p, seed, sa, sb)
        ctx.save_for_backward(h, Wab, ab, wcv)
        ctx.cfg = (p, seed, sa, sb, N, D, wc.shape)
        return s

    @staticmethod
    def backward(ctx, ds):
        h, Wab, ab, wcv = ctx.saved_tensors
        p, seed, sa, sb, N, D, wcshape = ctx.cfg
        dG, dwc, dbc, dbias = gate_bwd(ab, ds.contiguous(), wcv, N, D, p, seed, sa, sb)
        dh = gemm(dG, Wab, True, False, N, D, 2 * D) if ctx.needs_input_grad[0] else None
        dWab = gemm(dG, h, False, False, 2 * D, D, N)
        return (dh, dWab[:D], dbias[:D], dWab[D:], dbias[D:], dwc.reshape(wcshape), dbc, None, None, None, None)


def gate_scores(h, Wa, ba, Wb, bb, wc, bc, p=0.0, rng=None, tag=""):
    sa = sb = 0
    seed = None
    if p > 0.0:
        rng = rng or default_rng(h.device)
        sa = rng.site(tag + "att_a", tuple(h.shape), p)
        sb = rng.site(tag + "att_b", tuple(h.shape), p)
        seed = rng.seed
    return GateScoreFn.apply(h, Wa, ba, Wb, bb, wc, bc, float(p), seed, sa, sb)


class MhaFn(torch.autograd.Function):
    """Self-attention core of the ESAT layer (nn.MultiheadAttention inside nn.TransformerEncoderLayer, reference
    model/backbone_utils.py:113-127): packed qkv[L_total, 3d] of a slab of bags (`seg`; None = one bag) -> O[L_total, d]; attention
    never crosses a bag. The kernels read qkv as its two bf16x3 operand planes (`planes`: emitted by the in-projection's epilogue,
    else one advmil_split_planes pass here). ONE fused launch (advmil_mha_fwd: QK^T, online softmax, dropout, PV on the matrix
    pipe; no [H, L, L] tensor exists), three for the backward (advmil_mha_bwd1: prep, ONE pass over the scores for dQ / dK / dV, the
    reduce of dQ's per-key-block partial slabs; ADVMIL_ATTN_BWD=two: advmil_mha_bwd, prep + dQ + dK/dV). Ragged bags and any bag length are handled inside
    the kernels."""

    last_lse = None          # the log-sum-exp of the forward just run (side channel to ops.mha: the forward memo keeps the eval pass's)

    @staticmethod
    def forward(ctx, qkv, nhead, p, seed, sid, seg, rowoff, planes=None, lse_in=None):
        _chk(qkv, "qkv")
        qkv = qkv.contiguous()
        Lt, d3 = qkv.shape
        d = d3 // 3
        hd = d // nhead
        dev = qkv.device
        nseg = 1 if seg is None else seg.nseg
        mlen = Lt if seg is None else seg.max_len
        ptr = None if seg is None else seg.ptr
        shape_ = (Lt, nhead, hd, nseg)
        _stamp("b", "mha_fwd", shape_, 0.0)
        if planes is None:
            planes = split_planes(qkv)
        out = torch.empty(Lt, d, dtype=torch.float32, device=dev)
        if lse_in is not None and p > 0.0 and tuple(lse_in.shape) == (Lt, nhead):
            # the softmax statistics of the eval-mode pass over the same q | k | v (forward memo): only the dropout draw is new
            lse = lse_in
            _lib.check(_lib.lib().advmil_mha_fwd_lse(_p(planes.hi), _p(planes.lo), Lt, nhead, hd, nseg, _p(ptr), mlen, p, _p(seed), sid,
                                                     _p(rowoff), _p(out), _p(lse), _stream()), "mha_fwd_lse")
        else:
            lse = torch.empty(Lt, nhead, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().advmil_mha_fwd(_p(planes.hi), _p(planes.lo), Lt, nhead, hd, nseg, _p(ptr), mlen, p,
                                                 _p(seed if p > 0.0 else None), sid, _p(rowoff), _p(out), _p(lse), _stream()), "mha_fwd")
        MhaFn.last_lse = lse
        _stamp("e", "mha_fwd", shape_, 0.0)
        ctx.save_for_backward(planes.hi, planes.lo, out, lse)
        ctx.cfg = (nhead, hd, p, seed, sid, seg, rowoff)
        return out

    @staticmethod
    def backward(ctx, dO):
        qhi, qlo, out, lse = ctx.saved_tensors
        nhead, hd, p, seed, sid, seg, rowoff = ctx.cfg
        Lt = qhi.shape[0]
        L = _lib.lib()
        dO = dO.contiguous()
        dqkv = torch.empty(qhi.shape, dtype=torch.float32, device=qhi.device)
        nseg = 1 if seg is None else seg.nseg
        mlen = Lt if seg is None else seg.max_len
        ptr = None if seg is None else seg.ptr
        one = mha_bwd_single_pass(mlen, hd)
        wsb = L.advmil_mha_bwd1_workspace_bytes(Lt, nhead, hd, mlen) if one else L.advmil_mha_bwd_workspace_bytes(Lt, nhead, hd)
        ws = _ws(wsb, qhi.device)
        _stamp("b", "mha_bwd", (Lt, nhead, hd, nseg), 0.0)
        _lib.check((L.advmil_mha_bwd1 if one else L.advmil_mha_bwd)(
            _p(qhi), _p(qlo), _p(out), _p(dO), _p(lse), Lt, nhead, hd, nseg, _p(ptr), mlen, p, _p(seed if p > 0.0 else None), sid,
            _p(rowoff), _p(dqkv), _p(ws), wsb, _stream()), "mha_bwd1" if one else "mha_bwd")
        _stamp("e", "mha_bwd", (Lt, nhead, hd, nseg), 0.0)
        return dqkv, None, None, None, None, None, None, None, None


# Backward form of the attention core: "one" = advmil_mha_bwd1 (single pass over the scores, dQ through per-key-block partial
# slabs), "two" = advmil_mha_bwd (dQ and dK / dV launches, each recomputing the scores). ADVMIL_ATTN_BWD pins one of them.
ATTN_BWD = os.environ.get("ADVMIL_ATTN_BWD", "auto")


def mha_bwd_single_pass(max_len, head_dim):
    if ATTN_BWD in ("one", "two"):
        return ATTN_BWD == "one"
    return True


_MHA_P_WARNED = set()


def mha(qkv, nhead, p=0.0, rng=None, seg=None, rowoff=None):
    """qkv[L_total, 3d]; `seg` (ops.Segments) partitions the rows into bags. `rowoff`: optional int64 device tensor [nseg] added
    to each bag's local region rows to form the dropout stream's row ids (bag-parallel world-size invariance). Operand planes
    attached to qkv by the producing contraction (`_advmil_planes`) are used as they are."""
    sid, seed = 0, None
    if p > 0.0:
        # the kernels compare one hash byte per key against floor(256 p): p is quantised to 1/256 (0.25, the shipped rate, is exact)
        q = int(p * 256.0)
        if q == 0:
            raise ValueError(f"attention dropout p = {p} is below the kernels' resolution of 1/256 (it would silently be no dropout)")
        if abs(q / 256.0 - p) > 1e-9 and p not in _MHA_P_WARNED:
            _MHA_P_WARNED.add(p)
            import warnings
            warnings.warn(f"attention dropout p = {p} runs at {q}/256 = {q / 256.0:.6f} (byte-hash resolution of csrc/attn.hip)")
        rng = rng or default_rng(qkv.device)
        sid, seed = rng.site("mha_attn", (qkv.shape[0], nhead), p), rng.seed
    planes = getattr(qkv, "_advmil_planes", None) if qkv.is_contiguous() else None
    if getattr(qkv, "_advmil_planes_only", False) and planes is None:
        raise RuntimeError("advmil_amd: qkv was produced as operand planes only and lost them on the way to ops.mha")
    # Forward memo (MEMO_CHAIN): q | k | v of the train-mode pass ARE the eval-mode pass's (the in-projection's memoized output), so the
    # softmax statistics are too: the eval pass leaves its log-sum-exp, the train pass hands it to the kernel (advmil_mha_fwd_lse)
    memo = MEMO if (MHA_LSE_MEMO and MEMO_CHAIN and MEMO.mode is not None and planes is not None and qkv.data_ptr() in MEMO.derived) else None
    key = (("mha_lse", qkv.data_ptr(), tuple(qkv.shape), nhead, 1 if seg is None else seg.nseg, qkv.shape[0] if seg is None else seg.max_len)
           if memo is not None else None)
    lse_in = memo.store.pop(key, None) if (memo is not None and memo.mode == "replay" and p > 0.0) else None
    out = MhaFn.apply(qkv, nhead, float(p), seed, sid, seg, rowoff, planes, lse_in)
    lse, MhaFn.last_lse = MhaFn.last_lse, None
    if memo is not None and memo.mode == "record" and not torch.is_grad_enabled() and lse is not None:
        memo.store[key] = lse
    return out


MHA_LSE_MEMO = os.environ.get("ADVMIL_MHA_LSE_MEMO", "1") != "0"


class AddDropoutLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(o)): the two post-norm residuals of the ESAT layer (reference model/backbone_utils.py:113-127,
    nn.TransformerEncoderLayer with norm_first=False) as one launch each way; dropout index = row*d + col on stream `sid`."""

    @staticmethod
    def forward(ctx, x, o, gamma, beta, eps, p, seed, sid, rr=None):
        _chk(x, "x"); _chk(o, "o")
        x, o = x.contiguous(), o.contiguous()
        R, d = x.shape
        dev = x.device
        z = torch.empty_like(x)
        y = torch.empty_like(x)
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        g_, b_ = gamma.detach(), beta.detach()
        ypl = None
        AddDropoutLayerNormFn.last_planes = ypl
        _lib.check(_lib.lib().advmil_add_dropout_ln_fwd(_p(x), _p(o), _p(g_), _p(b_), eps, R, d, p, _p(seed if p > 0.0 else None), sid,
                                                        _p(rr if p > 0.0 else None), _p(z), _p(y), _p(mean), _p(rstd),
                                                        _p(None if ypl is None else ypl.hi), _p(None if ypl is None else ypl.lo), _stream()),
                   "add_dropout_ln_fwd")
        ctx.save_for_backward(z, g_, mean, rstd)
        ctx.cfg = (p, seed, sid, rr)
        gg, gb = _arena_grad(gamma), _arena_grad(beta)
        ctx.arena = (gg, gb) if (gg is not None and gb is not None) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        z, gamma, mean, rstd = ctx.saved_tensors
        p, seed, sid, rr = ctx.cfg
        R, d = z.shape
        L = _lib.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(z)
        do = torch.empty_like(z) if p > 0.0 else None
        acc = ctx.arena is not None
        dg = ctx.arena[0] if acc else torch.empty(d, dtype=torch.float32, device=z.device)
        db = ctx.arena[1] if acc else torch.empty(d, dtype=torch.float32, device=z.device)
        wsb = L.advmil_add_dropout_ln_bwd_workspace_bytes(R, d)
        ws = _ws(wsb, z.device)
        _lib.check(L.advmil_add_dropout_ln_bwd(_p(dy), _p(z), _p(gamma), _p(mean), _p(rstd), R, d, p, _p(seed if p > 0.0 else None),
                                               sid, _p(rr if p > 0.0 else None), _p(dx), _p(do), _p(dg), _p(db), 1 if acc else 0, _p(ws),
                                               wsb, _stream()), "add_dropout_ln_bwd")
        if do is None:
            do = dx
        return (dx, do, None, None, None, None, None, None, None) if acc else (dx, do, dg, db, None, None, None, None, None)


def add_dropout_layer_norm(x, o, gamma, beta, eps=1e-5, p=0.0, rng=None, tag=""):
    sid, seed, rr = 0, None, None
    if p > 0.0:
        rng = rng or default_rng(x.device)
        sid, seed, rr = rng.site(tag, tuple(o.shape), p), rng.seed, rng.row_map(o.shape[0], tag)
    y = AddDropoutLayerNormFn.apply(x, o, gamma, beta, float(eps), float(p), seed, sid, rr)
    ypl, AddDropoutLayerNormFn.last_planes = getattr(AddDropoutLayerNormFn, "last_planes", None), None
    if ypl is not None:
        y._advmil_planes = ypl
    return y


class SegMeanFn(torch.autograd.Function):
    """out[S, D] = Wn^T h with Wn[N, S] the normalised one-hot membership (per-cluster mean of DeepAttMISL,
    model/backbone.py:112-117) -- one contraction over the bag instead of 8 boolean gathers."""

    @staticmethod
    def forward(ctx, h, wn):
        N, D = h.shape
        S = wn.shape[1]
        ctx.save_for_backward(wn)
        ctx.dims = (N, D, S)
        return gemm(wn, h.contiguous(), False, False, S, D, N)

    @staticmethod
    def backward(ctx, dout):
        (wn,) = ctx.saved_tensors
        N, D, S = ctx.dims
        return gemm(wn, dout.contiguous(), True, False, N, D, S), None


class SegRowMeanFn(torch.autograd.Function):
    """Mean of the rows of h[N, D] per CONTIGUOUS segment of a slab (ops.Segments) -> [nseg, D]: the segmented pooling kernels
    with uniform weights (softmax of zero scores = 1/len per row); backward broadcasts dout[seg]/len to the rows."""

    @staticmethod
    def forward(ctx, h, seg):
        h = h.contiguous()
        N, D = h.shape
        A, pooled = softmax_pool(_const_zeros(N, h.device), h, N, D, seg)
        ctx.save_for_backward(A)
        ctx.seg = seg
        return pooled

    @staticmethod
    def backward(ctx, dout):
        (A,) = ctx.saved_tensors
        seg = ctx.seg
        dout = dout.contiguous()
        N, D = A.shape[0], dout.shape[1]
        dh = torch.empty(N, D, dtype=torch.float32, device=A.device)
        _lib.check(_lib.lib().advmil_seg_scale_rows(_p(dout), _p(A), _p(None if seg is None else seg.rowseg), N, D, _p(dh), _stream()),
                   "seg_scale_rows")
        return dh, None


def segmented_mean_rows(h, seg):
    return SegRowMeanFn.apply(h, seg)


def segmented_mean(h, seg_id, num_segments):
    """Mean of the rows of h[N, D] per segment id (float or int ids in [0, S)); empty segment -> zeros."""
    S = (num_segments + 3) // 4 * 4                       # contiguous dim of the [N, S] operand must be 4-aligned
    onehot = torch.nn.functional.one_hot(seg_id.reshape(-1).to(device=h.device, dtype=torch.long), S).to(h.dtype)
    wn = (onehot / onehot.sum(dim=0).clamp_min(1.0)).contiguous()
    return SegMeanFn.apply(h, wn)[:num_segments]


# ---------------------------------------------------------------------------------------
# PatchGCN pieces: LayerNorm+ReLU rows, graph CSR images, GENConv softmax aggregation
# ---------------------------------------------------------------------------------------
class LNReLUFn(torch.autograd.Function):
    """relu(LayerNorm(y)) per row (GENConv's norm='layer' MLP)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        _chk(y, "y")
        y = y.contiguous()
        N, d = y.shape
        out = torch.empty_like(y)
        mean = torch.empty(N, dtype=torch.float32, device=y.device)
        rstd = torch.empty(N, dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().advmil_ln_relu_fwd(_p(y), _p(gamma), _p(beta), eps, N, d, _p(out), _p(mean), _p(rstd), _stream()),
                   "ln_relu_fwd")
        ctx.save_for_backward(y, gamma.detach(), beta.detach(), mean, rstd)
        gg, gb = _arena_grad(gamma), _arena_grad(beta)
        ctx.arena = (gg, gb) if (gg is not None and gb is not None) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        N, d = y.shape
        L = _lib.lib()
        dy = torch.empty_like(y)
        acc = ctx.arena is not None
        dg = ctx.arena[0] if acc else torch.empty(d, dtype=torch.float32, device=y.device)
        db = ctx.arena[1] if acc else torch.empty(d, dtype=torch.float32, device=y.device)
        wsb = L.advmil_ln_relu_bwd_workspace_bytes(N, d)
        ws = _ws(wsb, y.device)
        _lib.check(L.advmil_ln_relu_bwd(_p(dout.contiguous()), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), N, d, _p(dy), _p(dg),
                                        _p(db), 1 if acc else 0, _p(ws), wsb, _stream()), "ln_relu_bwd")
        return (dy, None, None, None) if acc else (dy, dg, db, None)


def ln_relu(y, gamma, beta, eps=1e-5):
    return LNReLUFn.apply(y, gamma, beta, eps)


class GraphCSR:
    """Both CSR images of a WSI patch graph edge_index[2, E] (row 0 = source, row 1 = target; layout of
    tools/patchgcn_graph_s2.py:78-80). Built once per graph with integer device ops (sort / bincount / cumsum)."""

    def __init__(self, edge_index, num_nodes):
        ei = edge_index.to(torch.long)
        src, dst = ei[0], ei[1]
        self.N = int(num_nodes)
        self.E = int(src.numel())
        self.rowptr_dst, self.col_src = self._csr(dst, src)
        self.rowptr_src, self.col_dst = self._csr(src, dst)

    def _csr(self, key, val):
        order = torch.argsort(key, stable=True)
        counts = torch.bincount(key, minlength=self.N)
        rowptr = torch.zeros(self.N + 1, dtype=torch.int32, device=key.device)
        rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return rowptr.contiguous(), val[order].to(torch.int32).contiguous()


def graph_csr(data):
    """Cached GraphCSR of a graph object with `.edge_index` (and `.x`)."""
    g = getattr(data, "_advmil_csr", None)
    if g is None:
        g = GraphCSR(data.edge_index, data.x.shape[0])
        try:
            data._advmil_csr = g
        except Exception:
            pass
    return g


class GenConvAggFn(torch.autograd.Function):
    """out = softmax-aggregated messages + x (GENConv before its MLP); t is the learnable temperature [1]."""

    @staticmethod
    def forward(ctx, x, t, csr, eps, keep_rows=True):
        _chk(x, "x")
        x = x.contiguous()
        N, C = x.shape
        out = torch.empty_like(x)
        # (needs_input_grad reflects requires_grad whatever the grad mode: t is a Parameter, so a no-grad pass must be asked about itself)
        keep = keep_rows and any(ctx.needs_input_grad[:2])       # evaluation / no-grad passes: `out` is the only row written
        lse = torch.empty_like(x) if keep else None
        agg = torch.empty_like(x) if keep else None
        _lib.check(_lib.lib().advmil_genconv_fwd(_p(x), _p(csr.rowptr_dst), _p(csr.col_src), _p(t), eps, N, C, _p(out), _p(lse),
                                                 _p(agg), _stream()), "genconv_fwd")
        if keep:
            ctx.save_for_backward(x, t.detach(), lse, agg)
        ctx.csr, ctx.eps = csr, eps
        return out

    @staticmethod
    def backward(ctx, dout):
        x, t, lse, agg = ctx.saved_tensors
        csr = ctx.csr
        N, C = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dt = torch.empty(1, device=x.device, dtype=torch.float32)     # d/dt of the softmax weights: falls out of the same edge walk
        nws = _lib.lib().advmil_genconv_bwd_workspace_bytes(N, C)
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().advmil_genconv_bwd(_p(dout), _p(x), _p(agg), _p(lse), _p(csr.rowptr_src), _p(csr.col_dst), _p(t),
                                                 ctx.eps, N, C, _p(dx), _p(dt), _p(ws), nws, _stream()), "genconv_bwd")
        return dx, dt, None, None, None


def genconv_aggregate(x, t, csr, eps=1e-7):
    if csr.E == 0:                           # a graph without edges: nothing is aggregated (the kernels take no empty edge arrays)
        return x + 0.0 * t.sum()
    return GenConvAggFn.apply(x, t, csr, eps, torch.is_grad_enabled())


# ---------------------------------------------------------------------------------------
# the step's scalar losses: value + analytic gradient in one launch (advmil_gan_d_loss / advmil_gan_g_loss)
# ---------------------------------------------------------------------------------------
_WHICH = {"bce": 0, "hinge": 1, "wasserstein": 2}


class GanDLossFn(torch.autograd.Function):
    """loss = sum_i term_f(fake_i) / n_fake + sum_i mask_i term_r(real_i) / n_real; also returns sum mask*real and sum fake."""

    @staticmethod
    def forward(ctx, fake, real, mask, which, inv_nf, inv_nr, root=False):
        ctx.root = root
        fake = fake.contiguous().reshape(-1)
        nr = 0 if real is None else real.numel()
        real_c = None if real is None else real.contiguous().reshape(-1)
        out = torch.empty(3, dtype=torch.float32, device=fake.device)
        gf = torch.empty_like(fake)
        gr = None if real is None else torch.empty_like(real_c)
        _lib.check(_lib.lib().advmil_gan_d_loss(_p(fake), fake.numel(), _p(real_c), _p(mask), nr, which, inv_nf, inv_nr, _p(out), _p(gf),
                                                _p(gr), _stream()), "gan_d_loss")
        ctx.save_for_backward(gf, gr if gr is not None else gf)
        ctx.has_real = real is not None
        ctx.shapes = (fake.shape, None if real is None else real.shape)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # no zeros[3] for the statistics output in the backward (one fill launch per step)
        return out[0], out

    @staticmethod
    def backward(ctx, go, _):
        gf, gr = ctx.saved_tensors
        if go is None:
            return (None,) * 7
        if ctx.root:          # the caller backpropagates from this loss itself (upstream gradient 1): the analytic gradients as they are
            return gf, gr.reshape(ctx.shapes[1]) if ctx.has_real else None, None, None, None, None, None
        return gf * go, (gr * go).reshape(ctx.shapes[1]) if ctx.has_real else None, None, None, None, None, None


class GanDLossStackedFn(torch.autograd.Function):
    """GanDLossFn on ONE score vector f2 = [fake scores (nb) | real scores (nb)] (the stacked discriminator pass): its gradient leaves as
    one vector too, so autograd runs no slice backward (two zero-fills, two copies and an add per step)."""

    @staticmethod
    def forward(ctx, f2, nb, mask, which, inv_nf, inv_nr, root=False):
        ctx.root = root
        f2c = f2.contiguous().reshape(-1)
        out = torch.empty(3, dtype=torch.float32, device=f2.device)
        g2 = torch.empty_like(f2c)
        fake, real, gf, gr = f2c[:nb], f2c[nb:], g2[:nb], g2[nb:]
        _lib.check(_lib.lib().advmil_gan_d_loss(_p(fake), nb, _p(real), _p(mask), real.numel(), which, inv_nf, inv_nr, _p(out), _p(gf),
                                                _p(gr), _stream()), "gan_d_loss")
        ctx.save_for_backward(g2)
        ctx.shape = f2.shape
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0], out

    @staticmethod
    def backward(ctx, go, _):
        (g2,) = ctx.saved_tensors
        if go is None:
            return (None,) * 7
        return (g2 if ctx.root else g2 * go).reshape(ctx.shape), None, None, None, None, None, None


def gan_d_loss_stacked(f2, nb, real_mask, which, n_fake, n_real, root=False):
    """gan_d_loss(f2[:nb], f2[nb:], ...) without slicing f2 in autograd."""
    return GanDLossStackedFn.apply(f2, int(nb), real_mask, _WHICH[which], 1.0 / float(n_fake), (1.0 / float(n_real)) if n_real > 0 else 0.0,
                                   root)


def gan_d_loss(fake, real, real_mask, which, n_fake, n_real, root=False):
    """-> (loss [0-dim, differentiable], stats[3] = {loss, sum mask*real, sum fake}). root=True: the loss is what the caller calls
    backward on (upstream gradient 1), so the backward hands out the analytic gradients without two multiply launches."""
    return GanDLossFn.apply(fake, real, real_mask, _WHICH[which], 1.0 / float(n_fake), (1.0 / float(n_real)) if n_real > 0 else 0.0,
                            root)


class GanGLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, fake, t, e, vis, alpha, gamma, l2, coef, inv_nf, inv_nv, root=False):
        ctx.root = root
        shape = pred.shape
        pred_c, fake_c = pred.contiguous().reshape(-1), fake.contiguous().reshape(-1)
        out = torch.empty(3, dtype=torch.float32, device=pred.device)
        gp, gf = torch.empty_like(pred_c), torch.empty_like(fake_c)
        t_c, e_c = t.contiguous().reshape(-1), e.contiguous().reshape(-1)      # named: a temporary's block would be reused at once
        _lib.check(_lib.lib().advmil_gan_g_loss(_p(pred_c), _p(t_c), _p(e_c), _p(vis), _p(fake_c), pred_c.numel(), alpha, gamma, l2,
                                                coef, inv_nf, inv_nv, _p(out), _p(gp), _p(gf), _stream()), "gan_g_loss")
        ctx.save_for_backward(gp, gf)
        ctx.shapes = (shape, fake.shape)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0], out

    @staticmethod
    def backward(ctx, go, _):
        gp, gf = ctx.saved_tensors
        if go is None:
            return (None,) * 12
        if ctx.root:
            return (gp.reshape(ctx.shapes[0]), gf.reshape(ctx.shapes[1])) + (None,) * 10
        return ((gp * go).reshape(ctx.shapes[0]), (gf * go).reshape(ctx.shapes[1])) + (None,) * 10


def gan_g_loss(pred, fake, t, e, vis_mask, alpha, gamma, norm, coef, n_fake, n_vis, root=False):
    """-> (total [0-dim, differentiable], stats[3] = {total, reg, gen}); reg = 0 when no label is visible (n_vis == 0)."""
    return GanGLossFn.apply(pred, fake, t, e, vis_mask, float(alpha), float(gamma), 1 if norm == "l2" else 0, float(coef),
                            1.0 / float(n_fake), (1.0 / float(n_vis)) if n_vis > 0 else 0.0, root)


class SkinnyLinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for in_features == 1 or out_features == 1 (advmil_skinny_linear_fwd/bwd): one launch each way, weight
    and bias gradients accumulated straight into the optimizer arena when the parameters live there."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x = x.contiguous()
        W2 = W.detach().reshape(W.shape[0], -1).contiguous()
        B, K = x.shape
        N = W2.shape[0]
        y = torch.empty(B, N, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().advmil_skinny_linear_fwd(_p(x), _p(W2), _p(b), B, K, N, act, _p(y), _stream()), "skinny_linear_fwd")
        ctx.save_for_backward(x, W2, y)
        ctx.cfg = (act, B, K, N, W.shape, b is not None)
        ctx.gW, ctx.gb = _arena_grad(W), _arena_grad(b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2, y = ctx.saved_tensors
        act, B, K, N, wshape, has_b = ctx.cfg
        dy = dy.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]
        dx = torch.empty(B, K, dtype=torch.float32, device=dy.device) if need_x else None
        arena = need_w and ctx.gW is not None and (not need_b or ctx.gb is not None)
        if arena:
            dW, db = ctx.gW, (ctx.gb if need_b else None)
        else:
            dW = torch.empty(N, K, dtype=torch.float32, device=dy.device) if need_w else None
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if need_b else None
        _lib.check(_lib.lib().advmil_skinny_linear_bwd(_p(x), _p(W2), _p(y), _p(dy), B, K, N, act, _p(dx), _p(dW), _p(db),
                                                       1 if arena else 0, _stream()), "skinny_linear_bwd")
        if arena:
            return dx, None, None, None
        return dx, (None if dW is None else dW.reshape(wshape)), db, None


def skinny_linear(x, W, b, act="none"):
    return SkinnyLinearFn.apply(x, W, b, _ACT[act])


def gate_interleave(Wa, ba, Wb, bb, D, planes=False, pair32=False):
    """Rows a0, b0, a1, b1, ... of the two attention branches as one [2D, D] matrix (+ its bf16x3 planes) and the interleaved bias:
    the operand layout of the fused gate score (advmil_gate_interleave)."""
    Wa, Wb, ba, bb = (t.contiguous() for t in (Wa, Wb, ba, bb))
    dev = Wa.device
    Wi = torch.empty(2 * D, D, dtype=torch.float32, device=dev)
    bi = torch.empty(2 * D, dtype=torch.float32, device=dev)
    pl = Planes.alloc((2 * D, D), dev) if planes else None
    _lib.check(_lib.lib().advmil_gate_interleave(_p(Wa), _p(Wb), _p(ba), _p(bb), D, _p(Wi), _p(None if pl is None else pl.hi),
                                                 _p(None if pl is None else pl.lo), _p(bi), 1 if pair32 else 0, _stream()), "gate_interleave")
    return Wi, bi, pl


def gate_partial_sum(partial, bc=None):
    """s[n] = sum_j partial[n, j] (+ bc): the fused gate score's per-column-block partials -> scores (advmil_gate_partial_sum)."""
    partial = partial.contiguous()
    N, npart = partial.shape
    s = torch.empty(N, dtype=torch.float32, device=partial.device)
    _lib.check(_lib.lib().advmil_gate_partial_sum(_p(partial), npart, _p(None if bc is None else bc.reshape(-1)), N, _p(s), _stream()),
               "gate_partial_sum")
    return s


class PrjHeadFn(torch.autograd.Function):
    """out[B,1] = <u, t> + src w^T + bias (advmil_prj_head_fwd/bwd): the projection discriminator's head (reference GANSurv.py:96-105),
    one launch each way; w / bias gradients accumulated straight into the optimizer arena when the parameters live there."""

    @staticmethod
    def forward(ctx, u, t, src, W, b):
        u, t = u.contiguous(), t.contiguous()
        B, d = u.shape
        src_ = None if src is None else src.contiguous()
        w = None if W is None else W.detach().reshape(-1).contiguous()
        out = torch.empty(B, 1, dtype=torch.float32, device=u.device)
        _lib.check(_lib.lib().advmil_prj_head_fwd(_p(u), _p(t), _p(src_), _p(w), _p(None if b is None else b.detach()), B, d, _p(out),
                                                  _stream()), "prj_head_fwd")
        ctx.save_for_backward(u, t, src_, w)
        ctx.cfg = (B, d, None if W is None else W.shape, b is not None)
        ctx.gW, ctx.gb = _arena_grad(W), _arena_grad(b)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, t, src, w = ctx.saved_tensors
        B, d, wshape, has_b = ctx.cfg
        dout = dout.contiguous()
        nu, nt, ns, nw = ctx.needs_input_grad[0], ctx.needs_input_grad[1], src is not None and ctx.needs_input_grad[2], \
            src is not None and ctx.needs_input_grad[3]
        nb = src is not None and has_b and ctx.needs_input_grad[4]
        new = lambda: torch.empty(B, d, dtype=torch.float32, device=dout.device)  # noqa: E731
        du, dt, ds = (new() if nu else None), (new() if nt else None), (new() if ns else None)
        arena = (nw or nb) and (not nw or ctx.gW is not None) and (not nb or ctx.gb is not None)
        if arena:
            dw, db = (ctx.gW if nw else None), (ctx.gb if nb else None)
        else:
            dw = torch.empty(d, dtype=torch.float32, device=dout.device) if nw else None
            db = torch.empty(1, dtype=torch.float32, device=dout.device) if nb else None
        _lib.check(_lib.lib().advmil_prj_head_bwd(_p(dout), _p(u), _p(t), _p(src), _p(w), B, d, _p(du), _p(dt), _p(ds), _p(dw), _p(db),
                                                  1 if arena else 0, _stream()), "prj_head_bwd")
        if arena:
            return du, dt, ds, None, None
        return du, dt, ds, (None if dw is None else dw.reshape(wshape)), db


def prj_head(u, t, src=None, W=None, b=None):
    return PrjHeadFn.apply(u, t, src, W, b)


# ---------------------------------------------------------------------------------------
# the discriminator's bag-level tail as one launch each way (advmil_dtail_fwd / advmil_dtail_bwd, csrc/tail.hip)
# ---------------------------------------------------------------------------------------
DTAIL = os.environ.get("ADVMIL_DTAIL", "1") != "0"


class TailSpec:
    """Host description of one tail call: the two chains' layers [(W, bias, act code, p, stream id)], the projection layer, the dropout
    seed / row map. Parameters are passed to DTailFn.apply again, flattened, so that autograd sees them."""
    __slots__ = ("x", "y", "prj_w", "prj_b", "prj_src", "seed", "rr")

    def __init__(self, x, y, prj_w, prj_b, prj_src, seed, rr):
        self.x, self.y, self.prj_w, self.prj_b, self.prj_src, self.seed, self.rr = x, y, prj_w, prj_b, prj_src, seed, rr

    def params(self):
        out = []
        for (W, b, _, _, _) in self.x + self.y:
            out += [W, b]
        return out + [self.prj_w, self.prj_b]


def dtail_ok(B, spec):
    """Can this tail run as the fused launch? B <= 32 rows, widths <= 256, every trainable parameter with an arena slot (the kernel ADDS
    its weight gradients in place)."""
    if not DTAIL or B > 32 or not (1 <= len(spec.x) <= _lib.TAIL_MAXL) or not (1 <= len(spec.y) <= _lib.TAIL_MAXL):
        return False
    for (W, b, _, _, _) in spec.x + spec.y:
        if W.dim() != 2 or max(W.shape) > 256 or W.dtype != torch.float32 or not W.is_cuda or not W.is_contiguous():
            return False
    if spec.x[-1][0].shape[0] != spec.y[-1][0].shape[0]:
        return False
    for p in spec.params():
        if p is not None and p.requires_grad and _arena_grad(p) is None:
            return False
    return True


def _fill_tail(dt, B, spec, ys_x, ys_y, xin, tin, u, slots):
    """slots: None (forward), or the gradient-arena slot of every parameter in spec.params() order (None = not wanted), as they
    stood when the FORWARD ran -- a discriminator frozen for the generator update (requires_grad off around the forward) stays frozen
    in the backward even though the flags are back on by then."""
    dt.B, dt.nx, dt.ny, dt.prj_src = B, len(spec.x), len(spec.y), spec.prj_src
    dt.xin, dt.tin = xin.data_ptr(), tin.data_ptr()
    dt.u = None if u is None else u.data_ptr()
    k = 0
    for arr, layers, ys in ((dt.x, spec.x, ys_x), (dt.y, spec.y, ys_y)):
        for i, ((W, b, act, p, sid), yb) in enumerate(zip(layers, ys)):
            L = arr[i]
            L.W, L.bias = W.data_ptr(), (None if b is None else b.data_ptr())
            L.y, L.K, L.N, L.act, L.drop_p, L.stream_id = yb.data_ptr(), W.shape[1], W.shape[0], act, float(p), sid
            gW = slots[k] if slots is not None else None
            gb = slots[k + 1] if slots is not None else None
            k += 2
            L.dW, L.dbias = (None if gW is None else gW.data_ptr()), (None if gb is None else gb.data_ptr())
    if spec.prj_src:
        dt.w_prj = spec.prj_w.data_ptr()
        dt.b_prj = None if spec.prj_b is None else spec.prj_b.data_ptr()
        gw = slots[k] if slots is not None else None
        gb = slots[k + 1] if slots is not None else None
        dt.dw_prj, dt.db_prj = (None if gw is None else gw.data_ptr()), (None if gb is None else gb.data_ptr())
    dt.seed = None if spec.seed is None else spec.seed.data_ptr()
    dt.rng_row = None if spec.rr is None else spec.rr.data_ptr()


class DTailFn(torch.autograd.Function):
    """f[B, 1] = <u, hid_t> + prj(hid_x | hid_t), hid_x = x-chain(eb), hid_t = y-chain(t), u = im (region-level inner product) or hid_x."""

    @staticmethod
    def forward(ctx, eb, im, t, spec, *params):
        eb, t = eb.contiguous(), t.contiguous()
        im_ = None if im is None else im.contiguous()
        B, dev = eb.shape[0], eb.device
        ys_x = [torch.empty(B, W.shape[0], dtype=torch.float32, device=dev) for (W, _, _, _, _) in spec.x]
        ys_y = [torch.empty(B, W.shape[0], dtype=torch.float32, device=dev) for (W, _, _, _, _) in spec.y]
        out = torch.empty(B, 1, dtype=torch.float32, device=dev)
        dt = _lib.DTail()
        _fill_tail(dt, B, spec, ys_x, ys_y, eb, t, im_, None)
        dt.out = out.data_ptr()
        ctx.slots = [(_arena_grad(p) if (p is not None and ctx.needs_input_grad[4 + j]) else None) for j, p in enumerate(spec.params())]
        _lib.check(_lib.lib().advmil_dtail_fwd(ctypes.byref(dt), _stream()), "dtail_fwd")
        ctx.spec, ctx.B = spec, B
        ctx.nx = len(ys_x)
        ctx.has_im = im_ is not None
        ctx.save_for_backward(eb, t, *( [im_] if im_ is not None else []), *ys_x, *ys_y)
        return out

    @staticmethod
    def backward(ctx, dout):
        sv = ctx.saved_tensors
        eb, t = sv[0], sv[1]
        o = 2
        im_ = None
        if ctx.has_im:
            im_, o = sv[2], 3
        ys_x, ys_y = list(sv[o:o + ctx.nx]), list(sv[o + ctx.nx:])
        spec, B, dev = ctx.spec, ctx.B, eb.device
        dout = dout.contiguous()
        need_eb, need_im, need_t = ctx.needs_input_grad[0], (im_ is not None and ctx.needs_input_grad[1]), ctx.needs_input_grad[2]
        deb = torch.empty_like(eb) if need_eb else None
        dim_ = torch.empty_like(im_) if need_im else None
        dt_ = torch.empty_like(t) if need_t else None
        dt = _lib.DTail()
        _fill_tail(dt, B, spec, ys_x, ys_y, eb, t, im_, ctx.slots)
        dt.dout = dout.data_ptr()
        dt.dxin = None if deb is None else deb.data_ptr()
        dt.du = None if dim_ is None else dim_.data_ptr()
        dt.dtin = None if dt_ is None else dt_.data_ptr()
        _lib.check(_lib.lib().advmil_dtail_bwd(ctypes.byref(dt), _stream()), "dtail_bwd")
        return (deb, dim_, dt_, None) + (None,) * len(spec.params())


def dtail(eb, im, t, spec):
    return DTailFn.apply(eb, im, t, spec, *spec.params())


# ---------------------------------------------------------------------------------------
# the generator's bag-level head as two launches each way (advmil_ghead_fwd / _bwd, csrc/ghead.hip)
# ---------------------------------------------------------------------------------------
GHEAD = os.environ.get("ADVMIL_GHEAD", "1") != "0"


class GHeadSpec:
    """Host description of one head call: rho (Wr, br, p1, sid1) or None, MLPs[0] (W0, b0, p2, sid2), the output layer (W1, b1), the noise
    input (mode 0 none / 1 zeros / 2 the caller's tensor / 3 drawn in the kernel at site sid_noise), out_act (0 / 1 = sigmoid), the dropout
    seed and the bag-level row map."""
    __slots__ = ("Wr", "br", "p1", "sid1", "W0", "b0", "p2", "sid2", "W1", "b1", "noise_mode", "noise", "sid_noise", "out_act", "seed", "rr")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def params(self):
        return [self.Wr, self.br, self.W0, self.b0, self.W1, self.b1]


def ghead_ok(x, spec):
    """Can this head run as the fused launches? <= 32 bags, widths within the kernel's limits, every trainable parameter with an arena slot
    (the backward ADDS its gradients in place)."""
    if not GHEAD or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32 or not (1 <= x.shape[0] <= 32):
        return False
    d0 = x.shape[1]
    W0, W1, Wr = spec.W0, spec.W1, spec.Wr
    if W0 is None or W1 is None or W0.dim() != 2 or W1.dim() != 2 or W1.shape[0] != 1:
        return False
    d2 = W0.shape[0]
    d1 = 0 if Wr is None else Wr.shape[0]
    if Wr is not None and (Wr.dim() != 2 or Wr.shape[1] != d0 or W0.shape[1] != d1):
        return False
    if Wr is None and W0.shape[1] != d0:
        return False
    if d0 > 512 or d0 % 4 or d2 > 256 or d2 % 4 or (d1 if d1 else d2) % 16 or d1 > 1024:
        return False
    if W1.shape[1] != (d2 if spec.noise_mode == 0 else 2 * d2):
        return False
    for p in spec.params():
        if p is None:
            continue
        if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or (p.data_ptr() & 15 and p.dim() == 2 and p.shape[0] > 1):
            return False
        if p.requires_grad and torch.is_grad_enabled() and _arena_grad(p) is None:
            return False
    return True


def _fill_ghead(gh, x, spec, hs, h2, pred, ws, wsb):
    gh.B, gh.d0, gh.d2 = x.shape[0], x.shape[1], spec.W0.shape[0]
    gh.d1 = 0 if spec.Wr is None else spec.Wr.shape[0]
    gh.noise_mode, gh.out_act = spec.noise_mode, spec.out_act
    gh.x, gh.ldx = x.data_ptr(), x.stride(0)
    gh.Wr, gh.br = _p(spec.Wr), _p(spec.br)
    gh.W0, gh.b0, gh.W1, gh.b1 = _p(spec.W0), _p(spec.b0), _p(spec.W1), _p(spec.b1)
    gh.p1, gh.p2 = float(spec.p1 or 0.0), float(spec.p2 or 0.0)
    gh.seed = _p(spec.seed)
    gh.sid1, gh.sid2, gh.sid_noise = int(spec.sid1 or 0), int(spec.sid2 or 0), int(spec.sid_noise or 0)
    gh.rng_row = _p(spec.rr)
    gh.noise = _p(spec.noise)
    gh.hs, gh.h2, gh.pred = _p(hs), _p(h2), _p(pred)
    gh.ws, gh.ws_bytes = _p(ws), wsb


class GHeadFn(torch.autograd.Function):
    """pred [B, 1] = out_scale(MLPs[1](cat(MLPs[0](rho(x)), noise))) (Generator.finish behind the backbone's pooling)."""

    @staticmethod
    def forward(ctx, x, spec, pred_out, *params):
        if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() & 15:
            x = x.contiguous()
        B, dev = x.shape[0], x.device
        d2 = spec.W0.shape[0]
        d1 = 0 if spec.Wr is None else spec.Wr.shape[0]
        hs = torch.empty(B, d1 if d1 else d2, dtype=torch.float32, device=dev)
        h2 = torch.empty(B, d2, dtype=torch.float32, device=dev) if d1 else None
        if pred_out is not None and (tuple(pred_out.shape) != (B, 1) or pred_out.dtype != torch.float32 or not pred_out.is_contiguous()):
            pred_out = None
        pred = pred_out if pred_out is not None else torch.empty(B, 1, dtype=torch.float32, device=dev)
        L = _lib.lib()
        wsb = L.advmil_ghead_workspace_bytes(B, x.shape[1], d1, d2)
        ws = _ws(wsb, dev)
        gh = _lib.GHead()
        _fill_ghead(gh, x, spec, hs, h2, pred, ws, wsb)
        # (gradient slots as they stand NOW: parameters frozen around this forward stay frozen in the backward)
        ctx.slots = [(_arena_grad(p) if (p is not None and ctx.needs_input_grad[3 + j]) else None) for j, p in enumerate(spec.params())]
        _lib.check(L.advmil_ghead_fwd(ctypes.byref(gh), _stream()), "ghead_fwd")
        ctx.spec = spec
        ctx.has_h2 = h2 is not None
        ctx.save_for_backward(x, hs, pred, *([h2] if h2 is not None else []))
        return pred

    @staticmethod
    def backward(ctx, dpred):
        sv = ctx.saved_tensors
        x, hs, pred = sv[0], sv[1], sv[2]
        h2 = sv[3] if ctx.has_h2 else None
        spec, dev = ctx.spec, x.device
        dpred = dpred.contiguous()
        dx = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        L = _lib.lib()
        d1 = 0 if spec.Wr is None else spec.Wr.shape[0]
        wsb = L.advmil_ghead_workspace_bytes(x.shape[0], x.shape[1], d1, spec.W0.shape[0])
        ws = _ws(wsb, dev)
        gh = _lib.GHead()
        _fill_ghead(gh, x, spec, hs, h2, pred, ws, wsb)
        gh.dpred = dpred.data_ptr()
        gh.dx, gh.lddx = _p(dx), (x.shape[1] if dx is not None else 0)
        sl = ctx.slots
        gh.dWr, gh.dbr, gh.dW0, gh.db0, gh.dW1, gh.db1 = (_p(g) for g in sl)
        _lib.check(L.advmil_ghead_bwd(ctypes.byref(gh), _stream()), "ghead_bwd")
        return (dx, None, None) + (None,) * len(sl)


def ghead(x, spec, pred_out=None):
    """pred_out (no-grad calls only): write the predictions into this [B, 1] buffer instead of a fresh tensor."""
    return GHeadFn.apply(x, spec, pred_out, *spec.params())


# ---------------------------------------------------------------------------------------
# the discriminator's region-level network as one launch each way (advmil_dx_chain_fwd / _bwd, csrc/region.hip)
# ---------------------------------------------------------------------------------------
DX_CHAIN = os.environ.get("ADVMIL_DX_CHAIN", "1") != "0"


def _stacked_planes(Wa, Wb, D):
    """Planes of [Wa; Wb] ([2D, D]) as one strided view when the two weights lie side by side in the arena, else None."""
    pa_, pb_ = weight_planes(Wa), weight_planes(Wb)
    if pa_ is None or pb_ is None or not (_adjacent(pa_.hi, pb_.hi, 2) and _adjacent(pa_.lo, pb_.lo, 2)):
        return None
    return Planes(pa_.hi.as_strided((2 * D, D), (D, 1), pa_.hi.storage_offset()), pa_.lo.as_strided((2 * D, D), (D, 1), pa_.lo.storage_offset()))


def dx_chain_ok(e, W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc):
    """Can fc1 + the GAPool scorer over the region rows e [R, 128] run as the fused launches? bf16x3 arithmetic, the shipped widths
    (128 -> 64 -> 128, scorer 128 x 128), every weight with resident operand planes, gate branches side by side in the arena, and every
    trainable parameter with an arena slot (the backward ADDS into them)."""
    if not (DX_CHAIN and USE_PLANES and get_gemm_mode() == "bf16x3" and e.is_cuda and e.dim() == 2 and e.dtype == torch.float32
            and e.shape[1] == 128 and e.shape[0] >= 1):
        return False
    if tuple(W1.shape) != (64, 128) or tuple(W2.shape) != (128, 64) or tuple(Wa.shape) != (128, 128) or tuple(Wb.shape) != (128, 128):
        return False
    if any(t is None for t in (b1, b2, ba, bb, bc)) or wc.numel() != 128:
        return False
    if weight_planes(W1) is None or weight_planes(W2) is None or _stacked_planes(Wa, Wb, 128) is None or not _adjacent(ba, bb):
        return False
    ps = (W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc)
    if any(p.requires_grad and _arena_grad(p) is None for p in ps):
        return False
    gs = [_arena_grad(p) for p in (Wa, Wb, ba, bb)]
    if all(g is not None for g in gs) and not (_adjacent(gs[0], gs[1]) and _adjacent(gs[2], gs[3])):
        return False
    return True


class DxRegionPoolFn(torch.autograd.Function):
    """(pooled [B, 128], mean [B, 128] | None, A [R], fc [R, 128]) of the region rows e [R, 128]: fc = fc1(e), pooled = GAPool(fc) per bag,
    mean = per-bag mean of fc (EmbedXLayer / GAPool, reference model/model_utils.py:202-210, model/backbone_utils.py:47-56)."""

    @staticmethod
    def forward(ctx, e, W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc, p1, pg, seed, sids, rr, seg, want_mean, nograd):
        _chk(e, "e")
        e = e.contiguous()
        R, D = e.shape
        dev = e.device
        L = _lib.lib()
        w1pl, w2pl, wabpl = weight_planes(W1), weight_planes(W2), _stacked_planes(Wa, Wb, D)
        bab, _ = _stack2(ba, bb, D, 0)
        wcv = wc.detach().reshape(-1)
        keep = not nograd
        h1 = torch.empty(R, 64, dtype=torch.float32, device=dev) if keep else None
        ab = torch.empty(R, 2 * D, dtype=torch.float32, device=dev) if keep else None
        fc = torch.empty(R, D, dtype=torch.float32, device=dev)
        s = torch.empty(R, dtype=torch.float32, device=dev)
        use = seed is not None and (p1 > 0.0 or pg > 0.0)
        _lib.check(L.advmil_dx_chain_fwd(_p(e), R, D, _p(w1pl.hi), _p(w1pl.lo), _p(b1.detach()), _p(w2pl.hi), _p(w2pl.lo), _p(b2.detach()),
                                         _p(wabpl.hi), _p(wabpl.lo), _p(bab), _p(wcv), _p(bc.detach()), float(p1), float(pg),
                                         _p(seed if use else None), sids[0], sids[1], sids[2], _p(rr if use else None), _p(h1), _p(fc), _p(ab),
                                         _p(s), _stream()), "dx_chain_fwd")
        mean = None
        if want_mean:
            A, pooled, mean = softmax_pool_mean(s, fc, R, D, seg)      # the per-bag mean of fc from the pooling's own pass over it
        else:
            A, pooled = softmax_pool(s, fc, R, D, seg)
        if keep:
            ctx.save_for_backward(e, h1, fc, ab, A, wcv, W1.detach(), W2.detach(), _stack2(Wa, Wb, D, D)[0])
            ctx.cfg = (p1, pg, seed if use else None, sids, rr if use else None, seg, R, D)
            gs = [_arena_grad(t) for t in (W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc)]
            ctx.g = gs
        ctx.mark_non_differentiable(A)
        ctx.set_materialize_grads(False)
        return pooled, mean, A, fc

    @staticmethod
    def backward(ctx, dpooled, dmean, _dA, dfc_ext):
        e, h1, fc, ab, A, wcv, W1, W2, Wab = ctx.saved_tensors
        p1, pg, seed, sids, rr, seg, R, D = ctx.cfg
        gW1, gb1, gW2, gb2, gWa, gba, gWb, gbb, gwc, gbc = ctx.g
        dev = e.device
        L = _lib.lib()
        nseg = 1 if seg is None else seg.nseg
        dpooled = (torch.zeros(nseg, D, dtype=torch.float32, device=dev) if dpooled is None else dpooled.contiguous().reshape(nseg, D))
        ds = softmax_pool_bwd(dpooled, None, A, fc, R, D, seg)
        # the weights transposed, as operand planes (three tiny matrices: one launch)
        w1t, w2t, wabt = Planes.alloc((D, 64), dev), Planes.alloc((64, D), dev), Planes.alloc((D, 2 * D), dev)
        _lib.check(L.advmil_dx_chain_prep(_p(W1), _p(W2), _p(Wab), D, _p(w1t.hi), _p(w1t.lo), _p(w2t.hi), _p(w2t.lo), _p(wabt.hi), _p(wabt.lo),
                                          _stream()), "dx_chain_prep")
        need_e = ctx.needs_input_grad[0]
        dG = torch.empty(R, 2 * D, dtype=torch.float32, device=dev)
        dfc = torch.empty(R, D, dtype=torch.float32, device=dev)
        dpre = torch.empty(R, 64, dtype=torch.float32, device=dev)
        de = torch.empty(R, D, dtype=torch.float32, device=dev) if need_e else None
        frozen = gW1 is None          # (dx_chain_ok: every trainable parameter has its slot; none has one when D is frozen)
        if frozen:
            scratch = torch.zeros(3 * D + 8 + D + 64, dtype=torch.float32, device=dev)
            dwc, dbab, dbc, db2, db1 = scratch[:D], scratch[D:3 * D], scratch[3 * D:3 * D + 1], scratch[3 * D + 8:4 * D + 8], scratch[4 * D + 8:]
        else:
            dwc, dbc, db2, db1 = gwc.view(-1), gbc, gb2, gb1
            dbab = gba.as_strided((2 * D,), (1,), gba.storage_offset())
        wsb = L.advmil_dx_chain_bwd_workspace_bytes(R, D)
        ws = _ws(wsb, dev)
        dm = None if dmean is None else dmean.contiguous()
        ext = None if dfc_ext is None else dfc_ext.contiguous()
        _lib.check(L.advmil_dx_chain_bwd(R, D, _p(ds), _p(A), _p(dpooled), _p(dm), _p(None if seg is None else seg.rowseg),
                                         _p(None if seg is None else seg.ptr), _p(ext), _p(h1), _p(ab), _p(wcv), float(p1), float(pg), _p(seed),
                                         sids[1], sids[2], _p(rr), _p(wabt.hi), _p(wabt.lo), _p(w2t.hi), _p(w2t.lo), _p(w1t.hi), _p(w1t.lo),
                                         _p(dG), _p(dfc), _p(dpre), _p(de), _p(dwc), _p(dbab), _p(dbc), _p(db2), _p(db1), _p(ws), wsb, _stream()),
                   "dx_chain_bwd")
        if not frozen:
            gWab = gWa.as_strided((2 * D, D), (D, 1), gWa.storage_offset())
            if TN_GROUP:                                                                          # the three in one launch
                gemm_tn_group([(dG, fc, gWab, True), (dfc, h1, gW2.view(D, 64), True), (dpre, e, gW1.view(64, D), True)])
            else:
                gemm(dG, fc, False, False, 2 * D, D, R, out=gWab, ldc=D, accumulate=True)             # dWab += dG^T fc
                gemm(dfc, h1, False, False, D, 64, R, out=gW2.view(D, 64), ldc=64, accumulate=True)   # dW2  += dfc^T h1
                gemm(dpre, e, False, False, 64, D, R, out=gW1.view(64, D), ldc=D, accumulate=True)    # dW1  += dpre^T e
        return (de,) + (None,) * 18


def dx_region_pool(e, W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc, p1=0.0, pg=0.0, rng=None, seg=None, want_mean=False):
    """Fused fc1 + GAPool (+ per-bag mean of fc) over the region rows; the dropout call sites are drawn in the order and under the tags of
    the layer-by-layer path (dx_fc1, gapool_att_a, gapool_att_b)."""
    R = e.shape[0]
    sids, seed, rr = (0, 0, 0), None, None
    if p1 > 0.0 or pg > 0.0:
        rng = rng or default_rng(e.device)
        s1 = sa = sb = 0
        if p1 > 0.0:
            s1 = rng.site("dx_fc1", (R, 64), p1)
            rr = rng.row_map(R, "dx_fc1")
        if pg > 0.0:
            sa = rng.site("gapool_att_a", (R, 128), pg)
            sb = rng.site("gapool_att_b", (R, 128), pg)
            rg = rng.row_map(R, "gapool_att_a")
            if p1 > 0.0 and (rg is not rr) and not (rg is not None and rr is not None and rg.data_ptr() == rr.data_ptr()):
                raise RuntimeError("bag-parallel: the region layers of one pass must share their row map")
            rr = rg
        sids, seed = (s1, sa, sb), rng.seed
    ps = (W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc)
    nograd = not torch.is_grad_enabled() or not (e.requires_grad or any(t.requires_grad for t in ps))
    return DxRegionPoolFn.apply(e, W1, b1, W2, b2, Wa, ba, Wb, bb, wc, bc, float(p1), float(pg), seed, sids, rr, seg, bool(want_mean), nograd)
