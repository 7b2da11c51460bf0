"""Bag ingest for the step slab (replaces the synchronous pageable `.cuda()` per bag and per epoch at reference
model/model_handler.py:315, and the same in its evaluation loop, 598-643).

The loader hands over CPU tensors `x[1, N, 1024]`. `SlabStager` owns two (pinned host slab, device slab, device plane slab) sets.
  * A NEW bag is copied into the pinned slab (pageable source: a small pool of copy threads, `host_copy_rows`; a pinned source is
    DMA'd directly) and sent to the device slab with an async H2D copy on a dedicated copy stream; one launch behind it derives its
    bf16x3 operand planes (`advmil_stage_bag`, split form).
  * A bag SEEN BEFORE comes out of the device-resident `BagCache` (one per device, fp32 rows, LRU under a byte budget, shared by the
    training loop and the evaluation passes through `BagCacheView`s scoped by dataset object): one launch on the copy stream writes
    its rows into the slab and derives the planes on the way.
  * The bags of one optimizer step land BACK TO BACK in HBM -> the step's `[sum N, 1024]` matrix and its planes are zero-copy views;
    `pad_rows` appends zero rows up to whole 256-row tiles (a dummy bag the handlers drop after pooling).
  * While step t computes on set k, the host already stages step t+1 into set k^1 (PCIe / D2D overlap compute); a set is only
    rewritten after the step that read it has finished (event from the compute stream).
`step_batches` is the generator both handlers' loops and evaluations are built on.
"""
import os

import torch


def effective_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (a container with `cpu.max` = 16 CPUs on a
    256-thread host runs 128 OpenMP threads at the speed of 16, erratically)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


_COPY_POOL = None


def host_copy_rows(dst, src):
    """dst[rows, C] (pinned staging slab) <- src[rows, C] (pageable loader tensor) by a small pool of threads running numpy copies
    (the GIL is released inside them). torch's `copy_` hands a 33 MB copy to every OpenMP thread of the process: under a container
    CPU quota that took 5-100 ms for 16 bags from call to call; 8 plain threads take 5-7 ms (tools/probe/host_copy_threads.py:
    108 GB/s), which keeps a pageable loader at the PCIe rate. ADVMIL_INGEST_THREADS (default min(8, usable CPUs); 0 = torch copy_)."""
    global _COPY_POOL
    nt = int(os.environ.get("ADVMIL_INGEST_THREADS", min(8, effective_cpus())))
    rows = src.shape[0]
    if nt <= 1 or rows < 4 * nt or src.dtype != dst.dtype or not src.is_contiguous() or not dst.is_contiguous():
        dst.copy_(src)
        return
    import numpy as np
    if _COPY_POOL is None or _COPY_POOL._max_workers != nt:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(nt, thread_name_prefix="advmil-ingest")
    if src.dtype == torch.bfloat16:                              # (numpy has no bfloat16: copy the same bytes as int16)
        dst, src = dst.view(torch.int16), src.view(torch.int16)
    d, s_ = dst.numpy(), src.numpy()
    step = (rows + nt - 1) // nt
    futs = [_COPY_POOL.submit(np.copyto, d[r0:r0 + step], s_[r0:r0 + step]) for r0 in range(0, rows, step)]
    for f in futs:
        f.result()


# cached bags enter the step slab as operand planes only (no copy of their fp32 rows) when the step computes in bf16x3
PLANES_ONLY_STAGE = os.environ.get("ADVMIL_STAGE_PLANES_ONLY", "1") != "0"


class SlabStager:
    """`store`: dtype of the device slab. torch.bfloat16 = the x_storage 'bf16' mode: bags are kept in HBM as ONE bf16 plane (the
    step slab is its own operand plane: ops.is_bf16_slab); fp32 host bags cross PCIe as they are and are rounded once on the copy
    stream behind their H2D copy, bf16 host bags (a loader that stores bf16 features) are DMA'd straight into the slab."""

    def __init__(self, device, channels=1024, dtype=torch.float32, store=None):
        self.device = torch.device(device)
        self.channels = channels
        self.dtype = dtype
        self.store = dtype if store is None else store
        self.dev32 = [None, None]         # bf16 store fed by fp32 host bags: the H2D landing area of one batch (rounded into `dev`)
        # (a lower stream priority for the copy stream was measured in round 5: nothing -- profiles/r05_probe_staging.txt)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.host = [None, None]
        self.dev = [None, None]
        self.free_evt = [None, None]      # recorded on the compute stream when the step reading pair k is enqueued
        self.h2d_evt = [None, None]       # recorded on the copy stream when pair k's H2D copies are all enqueued
        self.pl = [None, None]            # bf16x3 operand planes of the device slabs (allocated with the first batch that needs them)
        self.k = 1
        self.rows = 0
        self.views = []
        self.planes_rows = 0              # rows [0, planes_rows) of the current batch have their planes in self.pl[k]
        self.pad = 0                      # zero rows behind the batch (pad_rows): the step slab is rows + pad rows long
        self.split = False                # derive every staged bag's bf16x3 operand planes on the copy stream (set per batch by begin())
        self.hint_rows = 0                # expected rows of a step batch (first allocation of the slabs; see _ensure)
        self.max_bag_rows = 0             # longest bag staged so far (expect(): ragged cohorts start a batch with any length)

    def _ensure(self, k, rows):
        cap = 0 if self.dev[k] is None else self.dev[k].shape[0]
        if rows <= cap:
            return
        # (whole 8 MB of plane rows: keeps the lo plane's 64 KB skew.) `hint_rows`: what the caller expects a step batch to need --
        # growing bag by bag re-allocated and re-copied the pinned slab ten times in a first epoch (1.3 s of its 2.1 s)
        new_cap = (max(rows, cap * 2, int(self.hint_rows), 1024) + 4095) // 4096 * 4096
        host = torch.empty(new_cap, self.channels, dtype=self.dtype).pin_memory()
        dev = torch.empty(new_cap, self.channels, dtype=self.store, device=self.device)
        if self.store != self.dtype:
            d32 = torch.empty(new_cap, self.channels, dtype=self.dtype, device=self.device)
            d32.record_stream(self.copy_stream)
            self.dev32[k] = d32                                  # (only written and read on the copy stream, inside one `add`)
        # the block comes from the COMPUTE stream's allocator pool: kernels already enqueued there may still be using it, and the
        # copy stream is about to write it -> order the copy stream behind them, and tell the allocator about the second user
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        dev.record_stream(self.copy_stream)
        if self.rows and self.dev[k] is not None:               # growing in the middle of a batch: keep what is staged
            self.copy_stream.synchronize()
            host[:self.rows].copy_(self.host[k][:self.rows])
            dev[:self.rows].copy_(self.dev[k][:self.rows])
            torch.cuda.current_stream(self.device).synchronize()
        self.host[k], self.dev[k] = host, dev
        self.views = [self.dev[k][a:b].unsqueeze(0) for a, b in self._spans]

    def _ensure_planes(self, k):
        """Plane slab of pair k, as many rows as its fp32 slab (one allocation, hi then lo: ops.Planes.alloc)."""
        from . import ops
        cap = self.dev[k].shape[0]
        old = self.pl[k]
        if old is not None and old.hi.shape[0] >= cap:
            return
        pl = ops.Planes.alloc((cap, self.channels), self.device)
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))     # (same allocator-pool ordering as in _ensure)
        pl.hi.record_stream(self.copy_stream)
        if old is not None and self.planes_rows:                                 # grown in the middle of a batch
            with torch.cuda.stream(self.copy_stream):
                pl.hi[:self.planes_rows].copy_(old.hi[:self.planes_rows])
                pl.lo[:self.planes_rows].copy_(old.lo[:self.planes_rows])
        self.pl[k] = pl

    def expect(self, nb, first_rows):
        """Size hint before a batch of `nb` bags whose first bag has `first_rows` rows: 1.25 x nb x the longest bag seen so far. (A
        hint from the first bag alone re-allocated the pinned slabs whenever a batch happened to start with a longer bag.)"""
        self.max_bag_rows = max(self.max_bag_rows, int(first_rows))
        self.hint_rows = max(self.hint_rows, int(1.25 * nb * self.max_bag_rows))

    def begin(self):
        """Start staging a new step batch into the other buffer pair."""
        from . import ops
        self.split = bool(ops.USE_PLANES and ops.get_gemm_mode() == "bf16x3" and self.device.type == "cuda"
                          and self.store == torch.float32)
        self.k ^= 1
        if self.h2d_evt[self.k] is not None:                     # the pinned slab / kept loader tensors of this pair may still be DMA sources
            self.h2d_evt[self.k].synchronize()
        self.rows = 0
        self.planes_rows = 0
        self.pad = 0
        self.views = []
        self._spans = []
        self._keep = []
        self._stale = []                      # (a, b, cached rows): spans whose fp32 rows were NOT copied (planes-only staging)
        self.stale = False
        if self.free_evt[self.k] is not None:                    # do not overwrite bags a running step still reads
            self.copy_stream.wait_event(self.free_evt[self.k])

    def add(self, x_cpu):
        """Stage one bag x[1, N, C] (CPU) -> device view [1, N, C] inside the slab (valid after `ready()`)."""
        x2 = x_cpu.reshape(-1, x_cpu.shape[-1])
        n = x2.shape[0]
        self.max_bag_rows = max(self.max_bag_rows, n)
        k = self.k
        self._spans.append((self.rows, self.rows + n))
        self._ensure(k, self.rows + n)
        a, b = self.rows, self.rows + n
        if x2.is_pinned():                                        # DataLoader(pin_memory=True): DMA straight from the loader's buffer
            src = x2
            self._keep.append(x_cpu)                              # keep the source alive until the copy has run
        else:
            host_copy_rows(self.host[k][a:b], x2)                 # pageable -> pinned, a few plain threads (see host_copy_rows)
            src = self.host[k][a:b]
        with torch.cuda.stream(self.copy_stream):
            if self.store == self.dtype:
                self.dev[k][a:b].copy_(src, non_blocking=True)
            else:                                                 # fp32 over PCIe, rounded to the bf16 slab behind the copy
                self.dev32[k][a:b].copy_(src, non_blocking=True)
        if self.store != self.dtype:
            from . import _lib
            C = self.channels
            _lib.check(_lib.lib().advmil_split_planes(self.dev32[k].data_ptr() + a * C * 4, n * C, self.dev[k].data_ptr() + a * C * 2, None,
                                                      self.copy_stream.cuda_stream), "split_planes(bf16 slab)")
        if self.split and self.planes_rows == a and self._split_ok(n):
            # the bag's operand planes behind its H2D copy, on the copy stream: the step slab's planes are complete when it is
            self._ensure_planes(k)
            from . import _lib
            C, dst, pl = self.channels, self.dev[k], self.pl[k]
            p0 = dst.data_ptr() + a * C * 4
            if _lib.lib().advmil_stage_bag(p0, p0, n * C * 4, pl.hi.data_ptr() + a * C * 2, None, pl.lo.data_ptr() + a * C * 2, None,
                                           n * C * 2, self.copy_stream.cuda_stream) == 0:
                self.planes_rows = b
        self.rows = b
        if len(self.views) < len(self._spans):
            self.views.append(self.dev[k][a:b].unsqueeze(0))
        return self.views[-1]

    def _split_ok(self, n):
        return self.dtype == torch.float32 and (n * self.channels) % 8 == 0

    def add_device(self, x_dev, planes=None, ready_evt=None):
        """Stage one bag that is ALREADY in HBM (the device-resident bag cache): device-to-device copies of its fp32 rows -- and of
        its operand planes, when it carries them -- into the slab on the copy stream, i.e. under the compute of the step before.
        `ready_evt`: event recorded on the compute stream behind the kernels that produced the cached tensors."""
        x2 = x_dev.reshape(-1, x_dev.shape[-1])
        n = x2.shape[0]
        self.max_bag_rows = max(self.max_bag_rows, n)
        k = self.k
        self._spans.append((self.rows, self.rows + n))
        self._ensure(k, self.rows + n)
        a, b = self.rows, self.rows + n
        if x2.dtype != self.store:
            raise TypeError(f"SlabStager(store={self.store}): a cached bag of dtype {x2.dtype} cannot enter this slab")
        derive = planes is None and self.split and self._split_ok(n)        # fp32-only cache entry: planes made on the way
        with_planes = (planes is not None or derive) and self.planes_rows == a
        if with_planes:
            self._ensure_planes(k)
        if ready_evt is not None:
            self.copy_stream.wait_event(ready_evt)
        x_dev.record_stream(self.copy_stream)         # a cache eviction must not hand the block back while this launch reads it
        if planes is not None:
            planes.hi.record_stream(self.copy_stream)
            planes.lo.record_stream(self.copy_stream)
        # ONE launch on the copy stream for the bag's fp32 rows and both planes (advmil_stage_bag); three torch copy_ calls inside a
        # stream context cost ~45 us of host time per bag, 16 bags per step
        from . import _lib
        C = self.channels
        dst = self.dev[k]
        esz = dst.element_size()
        pl = self.pl[k] if with_planes else None
        # Planes-only staging (bf16x3 arithmetic): every contraction that reads the step slab reads its operand PLANES, so a cached bag's
        # fp32 rows need not be copied at all -- the launch derives the planes straight from the cache entry (dst == src: no row copy) and
        # moves 8 instead of 12 bytes per element under the running step. The slab's fp32 rows of such spans are STALE: the batch is
        # flagged (`stale`, -> `_advmil_fp32_stale` on the step slab, ops.gemm then takes planes only); a batch that ends up without
        # complete planes or below the handler's plane threshold gets its rows after all (`_backfill`, from `ready`).
        only = bool(derive and pl is not None and PLANES_ONLY_STAGE)
        rc = _lib.lib().advmil_stage_bag(
            (x2.data_ptr() if only else dst.data_ptr() + a * C * esz), x2.data_ptr(), n * C * esz,
            None if pl is None else pl.hi.data_ptr() + a * C * 2, None if (pl is None or derive) else planes.hi.data_ptr(),
            None if pl is None else pl.lo.data_ptr() + a * C * 2, None if (pl is None or derive) else planes.lo.data_ptr(),
            0 if pl is None else n * C * 2, self.copy_stream.cuda_stream)
        if rc == 0 and only:
            self._stale.append((a, b, x2))
        if rc != 0:                                   # unaligned rows (channels not a multiple of 8): the general copies
            with torch.cuda.stream(self.copy_stream):
                dst[a:b].copy_(x2, non_blocking=True)
                if pl is not None and not derive:
                    pl.hi[a:b].copy_(planes.hi.reshape(n, -1), non_blocking=True)
                    pl.lo[a:b].copy_(planes.lo.reshape(n, -1), non_blocking=True)
                elif pl is not None:
                    pl = None                         # (no planes for this batch: the step splits the slab)
        if pl is not None:
            self.planes_rows = b
        self._keep.append(x_dev)
        self.rows = b
        if len(self.views) < len(self._spans):
            self.views.append(self.dev[k][a:b].unsqueeze(0))
        return self.views[-1]

    def pad_rows(self, multiple=256, min_rows=4096):
        """Zero rows behind the staged bags so that the step slab is a whole number of `multiple`-row tiles (call before `ready`).
        The slab kernels' fast forms -- the plane-fed LDS-DMA contractions, the two-layer launch, the streaming epilogue -- take whole
        256-row tiles only, and the rows of a real step batch are a multiple of 16, not of 256: without the pad a step costs 18 % more
        (tools/probe/epoch_host_profile.py with PROBE_ODD=1: 4.43 against 3.75 ms). The handler treats the pad as one more bag whose
        pooled row it drops; zero rows keep every activation of that bag finite and its gradient exactly zero. -> pad rows."""
        self.pad = 0
        if multiple <= 1 or self.rows < min_rows or self.rows % multiple == 0:
            return 0
        pad, k, a = (-self.rows) % multiple, self.k, self.rows
        self._ensure(k, a + pad)
        with torch.cuda.stream(self.copy_stream):
            self.dev[k][a:a + pad].zero_()
            if self.pl[k] is not None and self.planes_rows == a:
                self.pl[k].hi[a:a + pad].zero_()
                self.pl[k].lo[a:a + pad].zero_()
                self.planes_rows = a + pad
        self.pad = pad
        return pad

    def batch_planes(self):
        """Operand planes of the whole staged batch (rows [0, rows + pad)), when every bag brought its own; else None."""
        n = self.rows + self.pad
        if self.rows and self.planes_rows == n and self.pl[self.k] is not None:
            from . import ops
            return ops.Planes(self.pl[self.k].hi[:n], self.pl[self.k].lo[:n])
        return None

    def _backfill(self):
        """The fp32 rows of the spans staged as planes only, after all (this batch will be read as fp32 rows)."""
        with torch.cuda.stream(self.copy_stream):
            for a, b, x2 in self._stale:
                self.dev[self.k][a:b].copy_(x2, non_blocking=True)
        self._stale = []

    def ready(self, need_rows=False):
        """Make the compute stream wait for the staged copies; returns the per-bag device views of this batch. need_rows: the caller will
        read the fp32 rows (a step batch that is not one zero-copy slab): spans staged as planes only get their rows now."""
        if self._stale:
            from . import ops                 # (the handler attaches the batch's planes under exactly this rule: one predicate for both)
            if need_rows or self.batch_planes() is None or not ops.slab_takes_planes(self.rows + self.pad, self.dev[self.k].shape[1]):
                self._backfill()
        self.stale = bool(self._stale)
        evt = torch.cuda.Event()
        evt.record(self.copy_stream)
        self.h2d_evt[self.k] = evt
        torch.cuda.current_stream(self.device).wait_event(evt)
        return list(self.views)

    def release(self):
        """Call after the step that reads this batch has been enqueued on the compute stream."""
        evt = torch.cuda.Event()
        evt.record(torch.cuda.current_stream(self.device))
        self.free_evt[self.k] = evt


class BagCache:
    """Device-resident bags across epochs (replaces the per-bag, per-EPOCH `.cuda()` of reference
    model/model_handler.py:315 / dataset/PatchWSI.py:65-83): the first time a bag comes through the staging slab it is copied once
    more, device to device, into its own HBM allocation (fp32 rows: 4 B per element); from then on a step batch is assembled from
    the cached bags by ONE launch per bag on the copy stream, while the step before computes (SlabStager.add_device ->
    advmil_stage_bag), which writes the bag's rows into the staging slab and derives its two bf16x3 operand planes on the way -- no
    PCIe traffic, no per-step split, no gather on the compute stream. The default budget of 30 % of 288 GB holds ~2 500 bags of 8192 patches. One cache per
    device, shared by the training loop and the evaluation passes (keys: (scope, patient index), BagCacheView). LRU under a byte
    budget (with an admission rule for cohorts larger than the budget, see `put`); a bag that does not fit is simply not kept."""

    def __init__(self, device, budget_bytes, with_planes=None):
        from collections import OrderedDict
        self.device = torch.device(device)
        self.budget = int(budget_bytes)
        # fp32 rows only by default: the staging launch derives the operand planes on the way into the step slab (advmil_stage_bag's
        # split form: 12 bytes moved per element instead of 16, 4 bytes held instead of 8). ADVMIL_CACHE_PLANES=1 keeps the planes too.
        self.with_planes = (os.environ.get("ADVMIL_CACHE_PLANES", "0") == "1") if with_planes is None else with_planes
        self.entries = OrderedDict()      # key -> [x [1, N, C] fp32 device tensor (+ `_advmil_bag_planes`), bytes, tick of its last use, fingerprint]
        self.bytes = 0
        self.hits = self.misses = self.evictions = self.refused = self.mismatches = 0
        self.tick = 0                     # counts lookups: the clock of the admission rule in `put`

    def get(self, key, fingerprint=None):
        """The cached bag, or None. `fingerprint` (bag_fingerprint of the loader's host tensor): an entry whose shape or sampled
        values differ from what the loader handed over NOW is not this bag (another loader under the same scope, a dataset that
        changes its bags between visits) -- it is dropped and the lookup is a miss."""
        self.tick += 1
        ent = self.entries.get(key)
        if ent is None:
            self.misses += 1
            return None
        if fingerprint is not None and ent[3] is not None and ent[3] != fingerprint[:2]:
            self.bytes -= self.entries.pop(key)[1]
            self.mismatches += 1
            self.misses += 1
            return None
        self.entries.move_to_end(key)
        ent[2] = self.tick
        self.hits += 1
        return ent[0]

    def put(self, key, x_dev, fingerprint=None):
        """Keep a private copy of the staged bag `x_dev` [1, N, C] (a view into the staging slab, valid on the current stream)."""
        from . import ops
        if key in self.entries or self.budget <= 0:
            return
        planes = self.with_planes and ops.USE_PLANES and ops.get_gemm_mode() == "bf16x3" and x_dev.dtype == torch.float32
        nbytes = x_dev.numel() * (8 if planes else x_dev.element_size())      # (a bf16 slab's bags: 2 bytes per element)
        if nbytes > self.budget:
            return
        # Admission when full. An epoch visits every bag once, in a new order: plain LRU over a cohort larger than the budget evicts
        # exactly the bags the next epoch needs (measured: 4 GB budget, 9 GB cohort -> 2 % hits, and every miss also paid for its
        # insertion). So a bag only displaces entries that have gone UNUSED for a long time -- 16 x the number of resident bags in
        # lookups, i.e. bags of a dataset that is no longer iterated; otherwise the resident set stays as it is and the hit rate is
        # the resident fraction of the cohort (same run: 43 % hits).
        stale_after = 16 * max(len(self.entries), 64)
        while self.bytes + nbytes > self.budget and self.entries:
            k0 = next(iter(self.entries))
            if self.tick - self.entries[k0][2] <= stale_after:
                self.refused += 1
                return
            self.bytes -= self.entries.pop(k0)[1]
            self.evictions += 1
        x = x_dev.clone()
        if planes:
            x._advmil_bag_planes = ops.split_planes(x.view(-1, x.shape[-1]))
        x._advmil_ready = torch.cuda.Event()          # the copy stream that later reads this entry waits for it
        x._advmil_ready.record(torch.cuda.current_stream(self.device))
        self.entries[key] = [x, nbytes, self.tick, None if fingerprint is None else fingerprint[:2]]
        self.bytes += nbytes

    def set_budget(self, budget_bytes):
        """Change the byte budget (the last handler / evaluation to ask decides); evicts down to it at once."""
        self.budget = int(budget_bytes)
        while self.bytes > self.budget and self.entries:
            self.bytes -= self.entries.popitem(last=False)[1][1]
            self.evictions += 1

    def clear(self):
        self.entries.clear()
        self.bytes = 0

    def drop_scope(self, scope):
        """Forget every bag of one scope (its dataset object is gone)."""
        for k in [k for k in self.entries if isinstance(k, tuple) and len(k) == 2 and k[0] == scope]:
            self.bytes -= self.entries.pop(k)[1]

    def stats(self):
        return {"bags": len(self.entries), "gb": round(self.bytes / 1e9, 3), "hits": self.hits, "misses": self.misses,
                "evictions": self.evictions, "refused": self.refused, "mismatches": self.mismatches}


def fingerprint_positions(n):
    """Flat element indices of a bag of n elements that bag_fingerprint samples."""
    return (0, n // 3, n // 2, (2 * n) // 3, n - 1) if n else ()


def bag_fingerprint(x0):
    """Cheap identity of a host bag [1, N, C]: its shape and five sampled elements (a few microseconds; no pass over the bag). Kept
    with the cache entry and compared on every hit, so a key that now names a different bag cannot serve a stale one."""
    flat = x0.reshape(-1)
    return (tuple(x0.shape), tuple(float(flat[i]) for i in fingerprint_positions(flat.numel())))


class BagCacheView:
    """One loader's window onto the device's shared BagCache: keys are (scope, patient index), so the training loop, the
    per-epoch validation / test passes (MyHandler.test_model) and the k-fold loaders of `exec_semi_sl` share ONE byte budget and
    one LRU order instead of one budget each. Counts its own hits / misses."""

    def __init__(self, cache, scope):
        self.cache, self.scope = cache, scope
        self.hits = self.misses = 0
        self._ev0, self._rf0 = cache.evictions, cache.refused

    def get(self, key, fingerprint=None):
        x = self.cache.get((self.scope, key), fingerprint)
        if x is None:
            self.misses += 1
        else:
            self.hits += 1
        return x

    def put(self, key, x_dev, fingerprint=None):
        self.cache.put((self.scope, key), x_dev, fingerprint)

    def stats(self):
        mine = [e[1] for k, e in self.cache.entries.items() if isinstance(k, tuple) and len(k) == 2 and k[0] == self.scope]
        return {"bags": len(mine), "gb": round(sum(mine) / 1e9, 3), "hits": self.hits, "misses": self.misses,
                "evictions": self.cache.evictions - self._ev0, "refused": self.cache.refused - self._rf0}


_DEVICE_CACHES = {}
_DEVICE_STAGERS = {}
_SCOPE_TOKENS = {}            # id(dataset) -> (weakref, token): a dataset object's identity for as long as it lives
_NEXT_TOKEN = [0]


def new_scope_token():
    _NEXT_TOKEN[0] += 1
    return _NEXT_TOKEN[0]


DEFAULT_BUDGET_FRACTION = 0.30


def default_budget(device):
    """30 % of the device's memory (86 GB of 288: ~2 500 bags of 8192 patches) unless cfg['bag_cache_gb'] / ADVMIL_BAG_CACHE_GB say otherwise."""
    return DEFAULT_BUDGET_FRACTION * torch.cuda.get_device_properties(torch.device(device)).total_memory


def device_bag_cache(device, budget_bytes=None):
    """THE bag cache of a device (created on first use). `budget_bytes`: None keeps the current budget (default_budget at
    creation: 30 % of the device's memory); a handler passes its configured budget, the static evaluation pass never changes it."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    c = _DEVICE_CACHES.get(key)
    if c is None:
        if budget_bytes is None:
            budget_bytes = default_budget(device)
        c = _DEVICE_CACHES[key] = BagCache(device, budget_bytes)
    elif budget_bytes is not None and int(budget_bytes) != c.budget:
        c.set_budget(budget_bytes)
    return c


def x_store_dtype(x_storage, host_dtype=torch.float32):
    """Device dtype of the bags for a configured x_storage ('fp32' | 'bf16' | None = ADVMIL_X_STORAGE, default fp32)."""
    xs = x_storage or os.environ.get("ADVMIL_X_STORAGE", "fp32")
    if xs not in ("fp32", "bf16"):
        raise ValueError(f"x_storage must be 'fp32' or 'bf16', got {xs!r}")
    return torch.bfloat16 if (xs == "bf16" or host_dtype == torch.bfloat16) else torch.float32


def device_stager(device, channels, dtype=torch.float32, store=None):
    """The evaluation passes' staging slabs (one SlabStager per device, bag width and dtypes; the training loop keeps its own)."""
    device = torch.device(device)
    store = dtype if store is None else store
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), int(channels), dtype, store)
    st = _DEVICE_STAGERS.get(key)
    if st is None:
        st = _DEVICE_STAGERS[key] = SlabStager(device, channels, dtype, store)
    return st


def dataset_scope(loader):
    """Cache scope of a loader's DATASET object -- `dataset[i]` is the same bag whichever DataLoader wraps it (the training loader
    and a train-set evaluation share their bags); None when the loader has no such object; False when its bags must not be kept:
    the reference's WSIPatch with `ratio_mask` draws a fresh random instance mask in every `__getitem__` (dataset/PatchWSI.py:73-74).
    Dies with the dataset: a weak reference drops its bags from every device cache."""
    import weakref
    ds = getattr(loader, "dataset", None)
    if ds is None:
        return None
    if getattr(ds, "ratio_mask", None) or getattr(ds, "advmil_no_cache", False):
        return False                                  # never cache: every visit is a different bag (or the dataset opted out)
    ent = _SCOPE_TOKENS.get(id(ds))
    if ent is not None and ent[0]() is ds:
        return ("ds", ent[1])
    tok = new_scope_token()

    def gone(_ref, _id=id(ds), _tok=tok):
        _SCOPE_TOKENS.pop(_id, None)
        for c in _DEVICE_CACHES.values():
            c.drop_scope(("ds", _tok))
    try:
        ref = weakref.ref(ds, gone)
    except TypeError:
        return None
    _SCOPE_TOKENS[id(ds)] = (ref, tok)
    return ("ds", tok)


def loader_cache_view(device, loader, fallback_scope=None, budget_bytes=None):
    """This loader's BagCacheView on the device's cache, or None (CPU device, caching off, randomly masked dataset, or no scope)."""
    import os
    device = torch.device(device)
    gb = os.environ.get("ADVMIL_BAG_CACHE_GB")
    if device.type != "cuda" or (gb is not None and float(gb) <= 0):
        return None
    scope = dataset_scope(loader)
    if scope is None:
        scope = fallback_scope
    if not scope:
        return None
    if gb is not None:
        budget_bytes = float(gb) * 1e9
    return BagCacheView(device_bag_cache(device, budget_bytes), scope)


class StepBatch:
    """`n` bags of a loader, in loader order. staged: their rows sit back to back in the staging slab (xs[j][0] are views of it; a zero-copy
    step slab, with `_advmil_stager_planes` on the first view when every bag came out of the cache with its operand planes)."""
    __slots__ = ("pos", "idx", "xs", "ys", "staged", "pad")

    def __init__(self, pos, idx, xs, ys, staged, pad=0):
        self.pos, self.idx, self.xs, self.ys, self.staged, self.pad = pos, idx, xs, ys, staged, pad   # pad: zero rows behind the bags


def step_batches(loader, device, nb, cache=None, stager=None, drop_last=False, stageable=None, group_unstaged=False, pad_multiple=0,
                 x_storage=None):
    """Walk `loader` ((idx, [x, ext], y) items, x[1, N, C] on the host) in step batches of `nb` bags whose rows are contiguous in HBM:
    host bags through the pinned staging slab on the copy stream, bags seen before out of the device-resident cache (device-to-
    device, on the copy stream too). Items `stageable(x)` rejects (device tensors, graphs, odd shapes) come as single-bag batches with
    staged = False, in order (`group_unstaged`: grouped `nb` at a time as well, for a training loop whose loader hands over device
    tensors). The consumer must have ENQUEUED its work on a batch before asking for the next one: the code behind
    the `yield` then keeps first-seen bags in the cache and lets the staging slab be rewritten once that work has run."""
    device = torch.device(device)
    pos, idxs, xs, ys, fresh = [], [], [], [], []
    loose = StepBatch([], [], [], [], False)      # unstaged items being grouped (group_unstaged)
    own = stager

    def ok(x0):
        return (torch.is_tensor(x0) and not x0.is_cuda and x0.dim() == 3 and x0.shape[0] == 1 and x0.shape[1] > 0
                and x0.dtype in (torch.float32, torch.bfloat16) and (stageable is None or stageable(x0)))

    def finish():
        pad = own.pad_rows(pad_multiple) if pad_multiple else 0       # (whole 256-row tiles: SlabStager.pad_rows)
        for j, v in enumerate(own.ready()):
            xs[j][0] = v
        bpl = own.batch_planes()
        if bpl is not None:
            xs[0][0]._advmil_stager_planes = bpl
            if own.stale:
                xs[0][0]._advmil_fp32_stale = True       # cached bags were staged as operand planes only: the slab's fp32 rows are not valid
        return StepBatch(list(pos), list(idxs), list(xs), list(ys), True, pad)

    def after():
        for key, j, fp in fresh:
            cache.put(key, xs[j][0], fp)
        own.release()
        del pos[:], idxs[:], xs[:], ys[:], fresh[:]

    for b, (idx, x, y) in enumerate(loader):
        x0 = x[0]
        if nb <= 1 or not ok(x0):
            if xs:
                if drop_last and group_unstaged:  # (a step batch is `nb` bags of ONE kind; a training loop never mixes them)
                    raise ValueError("step_batches: host and device bags alternate inside one step batch")
                yield finish()
                after()
            if group_unstaged and nb > 1:
                loose.pos.append(b); loose.idx.append(idx); loose.xs.append(list(x)); loose.ys.append(y)
                if len(loose.xs) == nb:
                    yield loose
                    loose = StepBatch([], [], [], [], False)
            else:
                yield StepBatch([b], [idx], [list(x)], [y], False)
            continue
        if loose.xs:
            raise ValueError("step_batches: host and device bags alternate inside one step batch")
        if own is None:
            own = device_stager(device, x0.shape[-1], x0.dtype, x_store_dtype(x_storage, x0.dtype))
        if not xs:
            own.expect(nb, x0.shape[1])
            own.begin()
        key = int(idx.reshape(-1)[0]) if cache is not None else None
        fp = bag_fingerprint(x0) if cache is not None else None
        hit = cache.get(key, fp) if cache is not None else None
        if hit is not None and hit.dtype != own.store:           # kept under another x_storage: not this slab's bag
            hit = None
        if hit is not None:
            # the entry keeps its event: ANY copy stream that reads it (the handler's stager, the evaluation's) is ordered behind
            # the kernels that made it; waiting on a completed event costs nothing
            v = own.add_device(hit, hit.__dict__.get("_advmil_bag_planes"), hit.__dict__.get("_advmil_ready"))
        else:
            if cache is not None:
                fresh.append((key, len(xs), fp))
            v = own.add(x0)
        pos.append(b); idxs.append(idx); xs.append([v] + list(x[1:])); ys.append(y)
        if len(xs) == nb:
            yield finish()
            after()
    if xs and not drop_last:
        yield finish()
        after()
    elif xs:
        own.release()
    if loose.xs and not drop_last:
        yield loose
