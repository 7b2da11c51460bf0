"""HIP-graph capture of the G+D optimizer step for a fixed group of resident bags.

The step is launch-bound when driven eagerly (hundreds of short kernels per bag), so the whole schedule --
D forward/backward over the group, D Adam, G forward/backward, G Adam, RNG seed advance -- is captured once
with torch.cuda.graph (hipGraph underneath) and replayed. The ctypes launches go to the capturing stream, so
the hand-written kernels are captured exactly like torch's own. Dropout/noise stay fresh across replays because
the kernels read the step seed from device memory and the graph itself bumps it.

Single process: ONE graph (optionally with the generator's training forward as a parallel branch: MyHandler.overlap_gfwd, measured
slower, off by default).

Under bag-parallel (world > 1) the collectives stay OUTSIDE the graphs (four segments: D backward | G backbone forward | D Adam + G
loss/backward | G Adam), so capture never depends on RCCL's graph support; D's all-reduce is started asynchronously before the
G-forward segment and waited for after it, so the exchange runs under that segment's kernels.
"""
import torch


class GraphedStep:
    def __init__(self, handler, xs, ys, ys_host, mode="wlabel", label_visible_mask=None, warmup=2, force_segments=False):
        self.force_segments = force_segments
        self.h = handler
        if handler.dp.world > 1 or force_segments:
            handler.overlap_gfwd = False     # the segments are separate graphs: a fork event cannot cross from one capture into another
        self.xs, self.ys = xs, ys
        self.plan = handler._plan(xs, ys, mode, label_visible_mask, ys_host)   # python ints: baked into the graph
        self.lrs = self._lrs()
        self.segments = []
        self.logs = None
        self._capture(warmup)

    def _lrs(self):
        return (self.h.optimizerG.param_groups[0]["lr"], self.h.optimizerD.param_groups[0]["lr"])

    # the step, cut at the collectives
    def _seg_disc(self):
        self.preds, self.fakes = self.h._disc_backward(0, self.xs, self.ys, self.plan)

    def _seg_gfwd(self):
        self.h._gen_forward(self.xs, self.plan)              # independent of D: overlaps D's gradient exchange

    def _seg_mid(self):
        self.h._log_d()                      # after the exchange: the logged statistics are the reduced ones
        self.h.optimizerD.step()
        self.h._gen_finish(0, self.xs, self.ys, self.plan)

    def _seg_end(self):
        self.h._log_g()
        self.h.optimizerG.step()
        self.h.rng.advance(1)

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
                self._eager()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        del h.history[:]                     # the warm-up steps' own logs are dry runs
        pool = torch.cuda.graph_pool_handle()
        if h.dp.world > 1 or self.force_segments:
            parts = (self._seg_disc, self._seg_gfwd, self._seg_mid, self._seg_end)
        else:
            parts = (lambda: (self._seg_disc(), self._seg_gfwd(), self._seg_mid(), self._seg_end()),)
        for fn in parts:
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                fn()
            self.segments.append(g)
        self.logs = list(h.history)                          # device scalars rewritten by every replay
        h.history[:] = saved
        self._st_d, self._st_g = h._st_d, h._st_g            # the statistics tensors this graph writes (reduced between segments)

    def replay(self):
        if self._lrs() != self.lrs:                          # lr is a launch constant: re-capture after a scheduler step
            self.segments, self.lrs = [], self._lrs()
            self._capture(0)
        segs = self.segments
        if len(segs) == 1:
            segs[0].replay()
            return
        segs[0].replay()
        self.h._st_d, self.h._st_g = self._st_d, self._st_g      # this graph's statistics tensors (another group may have run since)
        pend = self.h.dp.allreduce_async(self.h.optimizerD.flat_grad, self._st_d[0])     # D's exchange ...
        segs[1].replay()                                                                 # ... under the generator's backbone forward
        for w in pend:
            w.wait()
        segs[2].replay()
        self.h._reduce_g()
        segs[3].replay()
