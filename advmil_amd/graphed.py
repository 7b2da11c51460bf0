"""HIP-graph capture of the G+D optimizer step for a fixed group of resident bags.

The step is launch-bound when driven eagerly (hundreds of short kernels per bag), so the whole schedule --
D forward/backward over the group, D Adam, G forward/backward, G Adam, RNG seed advance -- is captured once
with torch.cuda.graph (hipGraph underneath) and replayed. The ctypes launches go to the capturing stream, so
the hand-written kernels are captured exactly like torch's own. Dropout/noise stay fresh across replays because
the kernels read the step seed from device memory and the graph itself bumps it.

Single process: ONE graph.

Under bag-parallel (world > 1) the collectives stay OUTSIDE the graphs by default (four segments: D backward | G backbone forward |
D Adam + G loss/backward | G Adam), so capture never depends on RCCL's graph support; D's all-reduce is started asynchronously before
the G-forward segment and waited for after it, so the exchange runs under that segment's kernels. `replay` stamps the two waits with
events (`exposed_allreduce_ms`): the time the compute stream stood still for an exchange.

ADVMIL_GRAPH_COLLECTIVES=1 (backend nccl = RCCL only): the two all-reduces are captured INSIDE the step graph -- a rank's step is then
one graph launch (on the launch-bound per-rank step of the strong split: three graph launches and two host-issued collectives less).
Every rank must take the same path, so the choice is agreed on with an all-reduce BEFORE any capture, a failed capture on any rank
sends all ranks back to the segments, and bench.py runs its multi-rank legs under a stall watchdog. Exercised on this image with a
one-rank RCCL communicator only (tests/test_parallel_gpu.py); the default stays the segment path until a node has run it.
"""
import os

import torch

from . import ops


# the two gradient-arena fills of a step folded into the Adam launches (ADVMIL_CLEAR_IN_ADAM=0: separate fill launches, for A/B timing)
CLEAR_IN_ADAM = os.environ.get("ADVMIL_CLEAR_IN_ADAM", "1") != "0"


class GraphedStep:
    def __init__(self, handler, xs, ys, ys_host, mode="wlabel", label_visible_mask=None, warmup=2, force_segments=False,
                 capture_collectives=None, plan=None, keep_warmup=False, pool=None, site_base=None):
        """plan: a step plan the caller built (the epoch loop's shape-keyed graphs hand over a plan whose device arrays are STATIC and
        rewritten per batch: model_handler.StaticStepPlan). keep_warmup: the warm-up steps are real training steps of the caller --
        their logs stay in handler.history and the last one's predictions / scores are kept in `warm_out`."""
        self.keep_warmup, self.warm_out = bool(keep_warmup), None
        self.pool = pool                     # a graph memory pool shared with other step graphs of the handler (they never run concurrently)
        # every pass over the step (warm-up, capture) numbers its dropout / noise call sites from here: the epoch loop's eager steps do
        # the same, so a replayed step draws exactly what the eager step it stands for would have drawn (same seed, same sites)
        self.site_base = site_base
        self.force_segments = force_segments
        self.h = handler
        self.wait_events = []                # (start, stop) event pairs around the two exchange waits of the last segmented replay
        self.captured_collectives = False
        if capture_collectives is None:
            # default ON: taken only when the backend is nccl (= RCCL) and every rank agrees (below); ADVMIL_GRAPH_COLLECTIVES=0 keeps the segments
            capture_collectives = os.environ.get("ADVMIL_GRAPH_COLLECTIVES", "1") != "0"
        dp = handler.dp
        want = bool(capture_collectives and dp.enabled and (dp.world > 1 or getattr(dp, "force", False)) and not force_segments
                    and torch.distributed.get_backend(dp.group) == "nccl")
        self._want_captured = self._agree(want)
        self.xs, self.ys = xs, ys
        self.plan = plan if plan is not None else handler._plan(xs, ys, mode, label_visible_mask, ys_host)   # python ints: baked into the graph
        self.lrs = self._lrs()
        self.segments = []
        self.logs = None
        self._capture(warmup)

    def _agree(self, flag):
        """True only if EVERY rank says so (one tiny MIN all-reduce, outside any capture)."""
        dp = self.h.dp
        if not (dp.enabled and dp.world > 1):
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=self.h.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=dp.group)
        return bool(float(t.item()) > 0.5)

    def _whole(self):
        """The step with its exchanges in stream order (what a single captured graph holds)."""
        self._seg_disc()
        self.h._reduce_d()
        self._seg_gfwd()
        self._seg_mid()
        self.h._reduce_g()
        self._seg_end()

    def _lrs(self):
        return (self.h.optimizerG.param_groups[0]["lr"], self.h.optimizerD.param_groups[0]["lr"])

    # the step, cut at the collectives
    def _seg_disc(self):
        self.preds, self.fakes = self.h._disc_backward(0, self.xs, self.ys, self.plan)

    def _seg_gfwd(self):
        self.h._gen_forward(self.xs, self.plan)              # independent of D: overlaps D's gradient exchange

    def _seg_mid(self):
        self.h._log_d()                      # after the exchange: the logged statistics are the reduced ones
        # (its counter is ticked with G's and the RNG seed at the end of the step: _seg_end; the gradient arena is cleared behind the read)
        self.h.optimizerD.step(tick=False, clear_grad=CLEAR_IN_ADAM)
        self.h._gen_finish(0, self.xs, self.ys, self.plan)

    def _seg_end(self):
        self.h._log_g()
        self.h.optimizerG.step(tick=False, abs_partial=self.h._abs_partial, clear_grad=CLEAR_IN_ADAM)
        ops.step_seed_tick(self.h.optimizerG.step_t, self.h.rng.seed, 1, step2=self.h.optimizerD.step_t)   # both step counters and the RNG seed: one launch

    def _eager(self):
        self._seg_disc()
        pend = self.h.dp.allreduce_async(self.h.optimizerD.flat_grad, self.h._st_d[0])
        self._seg_gfwd()
        for w in pend:
            w.wait()
        self._seg_mid()
        self.h._reduce_g()
        self._seg_end()

    def _capture(self, warmup):
        h = self.h
        saved = list(h.history)              # the caller's pending eager logs survive warm-up and capture (also on a re-capture)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                       # warm-up off the default stream, as capture requires
            for _ in range(warmup):
                if self.site_base is not None:
                    h.rng.counter = self.site_base
                self._eager()
            if self.keep_warmup and warmup:
                self.warm_out = (torch.cat(self.preds, dim=0).detach().clone(), torch.cat(self.fakes, dim=0).detach().clone())
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if self.keep_warmup:
            saved = list(h.history)          # (the caller's steps: their logs stay)
        del h.history[:]                     # the warm-up steps' own logs are dry runs
        pool = self.pool if self.pool is not None else torch.cuda.graph_pool_handle()
        for opt in (h.optimizerD, h.optimizerG):             # a captured zero_grad() launches no fill: replay() hands the graph clean arenas
            opt.capture_assumes_clean = CLEAR_IN_ADAM
        try:
            self._capture_graphs(pool)
        finally:
            for opt in (h.optimizerD, h.optimizerG):
                opt.capture_assumes_clean = False
        self.logs = list(h.history)                          # device scalars rewritten by every replay
        h.history[:] = saved
        self._st_d, self._st_g = h._st_d, h._st_g            # the statistics tensors this graph writes (reduced between segments)

    def _capture_graphs(self, pool):
        h = self.h
        if self.site_base is not None:
            h.rng.counter = self.site_base
        if self._want_captured:
            # the exchanges inside the graph: capture on every rank, then agree that it worked everywhere -- else all fall back together
            ok = True
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    self._whole()
            except Exception:
                ok = False
                torch.cuda.synchronize()
            if self._agree(ok):
                self.segments.append(g)
                self.captured_collectives = True
            else:
                self._want_captured = False
                del h.history[:]
        if not self.segments:
            if h.dp.world > 1 or self.force_segments or (h.dp.enabled and getattr(h.dp, "force", False)):
                parts = (self._seg_disc, self._seg_gfwd, self._seg_mid, self._seg_end)
            else:
                parts = (lambda: (self._seg_disc(), self._seg_gfwd(), self._seg_mid(), self._seg_end()),)
            for fn in parts:
                g = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread may query events while this thread captures
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    fn()
                self.segments.append(g)

    def replay(self):
        if self._lrs() != self.lrs:                          # lr is a launch constant: re-capture after a scheduler step
            self.segments, self.lrs = [], self._lrs()
            self._capture(0)
        segs = self.segments
        if CLEAR_IN_ADAM:
            # the captured step holds no fill launches: it starts from arenas its own Adam launches left clean. Anything else that wrote
            # them in between -- an eager step (which keeps its gradients readable), a kernel handed an arena slot by ops._arena_grad, a
            # torch-side write to some p.grad (version counter) -- leaves them dirty: clear them here, outside the graph
            for opt in (self.h.optimizerD, self.h.optimizerG):
                if not opt.grad_is_clean():
                    opt.flat_grad.zero_()
                opt.mark_grad_dirty()        # (the replay's backward launches dirty it; its Adam cleans it again: marked below)
        if len(segs) == 1:
            segs[0].replay()
            if CLEAR_IN_ADAM:
                self.h.optimizerD.mark_grad_clean(); self.h.optimizerG.mark_grad_clean()
            return
        segs[0].replay()
        self.h._st_d, self.h._st_g = self._st_d, self._st_g      # this graph's statistics tensors (another group may have run since)
        pend = self.h.dp.allreduce_async(self.h.optimizerD.flat_grad, self._st_d[0])     # D's exchange ...
        segs[1].replay()                                                                 # ... under the generator's backbone forward
        stamp = self.stamp_waits
        if stamp:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        for w in pend:
            w.wait()
        if stamp:
            ev[1].record()
        segs[2].replay()
        if stamp:
            ev[2].record()
        self.h._reduce_g()
        if stamp:
            ev[3].record()
            self.wait_events = [(ev[0], ev[1]), (ev[2], ev[3])]
        segs[3].replay()
        if CLEAR_IN_ADAM:
            self.h.optimizerD.mark_grad_clean(); self.h.optimizerG.mark_grad_clean()

    stamp_waits = False

    def exposed_allreduce_ms(self):
        """(D wait, G wait) of the last stamped segmented replay: how long the compute stream stood still for each exchange (after a
        synchronize). D's is what the overlap with the generator's forward did not hide; G's exchange has nothing to hide under."""
        return tuple(a.elapsed_time(b) for a, b in self.wait_events)
