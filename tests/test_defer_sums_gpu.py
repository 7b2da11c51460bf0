"""Deferred merges of parameter-gradient partials (include/advmil_hip.h::advmil_defer_sums, csrc/sumq.hip): one merge launch per backward
instead of one per partial set -- results must not move by a bit (the same partials are added in the same order)."""
import ctypes

import pytest
import torch

from advmil_amd import _lib, ops
from tests import helpers as H
from tests.test_parity_gpu import DEV, make_handler

pytestmark = pytest.mark.gpu


def _grads(kind, defer, nb=3, n=4352):
    old = ops.DEFER_SUMS
    ops.DEFER_SUMS = defer
    try:
        h, _, _ = make_handler(kind, bp_every_batch=nb)
        from advmil_amd import synth
        ext = (lambda i: H.T(synth.cluster_ids(0, i, n), DEV)) if kind == "cluster" else (lambda i: torch.zeros(1, 1, device=DEV))
        xs = [[H.bag(i, n, DEV), ext(i)] for i in range(nb)]
        ys_host = [H.label(i) for i in range(nb)]
        ys = [y.to(DEV) for y in ys_host]
        h.rng.reset(5)
        plan = h._plan(xs, ys, "wlabel", None, ys_host)
        h._disc_backward(0, xs, ys, plan)
        gd = h.optimizerD.flat_grad.clone()
        h._gen_backward(0, xs, ys, plan)
        gg = h.optimizerG.flat_grad.clone()
        torch.cuda.synchronize()
        return gd, gg
    finally:
        ops.DEFER_SUMS = old


@pytest.mark.parametrize("kind", ["abmil", "patch", "cluster"])
def test_deferred_merges_are_bitwise_neutral(kind):
    a = _grads(kind, False)
    b = _grads(kind, True)
    for x, y in zip(a, b):
        assert torch.isfinite(x).all()
        assert float(x.abs().max()) > 0
        assert torch.equal(x, y)


def test_merge_queue_counts_flushes_on_a_clash_and_leaves_plain_merges_alone():
    L = _lib.lib()
    s = ops._stream()
    x = torch.randn(4096, 64, device=DEV)
    out = torch.zeros(64, device=DEV)
    ref = x.double().sum(0)
    assert L.advmil_pending_sums(s) == -1
    with ops.deferred_sums():
        ops.colsum(x, 4096, 64, out=out)                  # accumulating: queued
        assert L.advmil_pending_sums(s) == 1
        assert float(out.abs().max()) == 0.0              # (nothing merged yet)
        fresh = ops.colsum(x, 4096, 64)                   # a plain (overwriting) merge is never deferred
        assert L.advmil_pending_sums(s) == 1
        assert float((fresh.double() - ref).abs().max()) < 1e-3
        ops.colsum(x, 4096, 64, out=out)                  # same destination again: the queued one goes first
        assert L.advmil_pending_sums(s) == 1
        assert float((out.double() - ref).abs().max()) < 1e-3
    assert L.advmil_pending_sums(s) == -1
    assert float((out.double() - 2 * ref).abs().max()) < 2e-3


def test_split_k_weight_gradient_goes_through_the_queue():
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        M, N, K = 128, 64, 16384
        a = torch.randn(K, M, device=DEV)
        b = torch.randn(K, N, device=DEV)
        _, sp = ops.gemm_plan(M, N, K, False, False)
        assert sp > 1
        base = torch.randn(M, N, device=DEV)
        o1, o2 = base.clone(), base.clone()
        ops.gemm(a, b, False, False, M, N, K, out=o1, ldc=N, accumulate=True)
        ops.ARENA_STORAGES.add(o2.untyped_storage().data_ptr())      # (ops.gemm defers folds into gradient arenas only: o2 plays one)
        try:
            with ops.deferred_sums():
                ops.gemm(a, b, False, False, M, N, K, out=o2, ldc=N, accumulate=True)
                assert _lib.lib().advmil_pending_sums(ops._stream()) == 1
                assert torch.equal(o2, base)
        finally:
            ops.ARENA_STORAGES.discard(o2.untyped_storage().data_ptr())
        assert torch.equal(o1, o2)
        ref = base.double() + a.double().t() @ b.double()
        assert float((o1.double() - ref).abs().max()) < 2e-2
    finally:
        ops.set_gemm_mode(prev)


def test_a_reader_inside_the_deferral_flushes_first():
    """The C-ABI contract of `accumulate` under deferral (advmil_hip.h): an accumulating split-K fold is postponed until the flush
    whatever the destination, so a caller that wants to read it before the deferral ends calls advmil_flush_sums -- after which it holds
    exactly what the undeferred call gives, and the queue is empty. (ops.gemm itself only lets folds into registered gradient arenas be
    deferred -- the buffer here is registered as one; a plain buffer: test_a_split_k_accumulate_into_a_plain_buffer_is_not_deferred.)"""
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        M, N, K = 128, 64, 16384
        a = torch.randn(K, M, device=DEV)
        b = torch.randn(K, N, device=DEV)
        base = torch.randn(M, N, device=DEV)
        o1, o2 = base.clone(), base.clone()
        ops.gemm(a, b, False, False, M, N, K, out=o1, ldc=N, accumulate=True)
        L, s = _lib.lib(), ops._stream()
        ops.ARENA_STORAGES.add(o2.untyped_storage().data_ptr())
        try:
            with ops.deferred_sums():
                ops.gemm(a, b, False, False, M, N, K, out=o2, ldc=N, accumulate=True)
                assert L.advmil_pending_sums(s) == 1 and torch.equal(o2, base)
                _lib.check(L.advmil_flush_sums(s), "flush_sums")
                assert L.advmil_pending_sums(s) == 0
                doubled = o2 * 2.0                               # a reader on the same stream, behind the flush
                assert torch.equal(o2, o1) and torch.equal(doubled, o1 * 2.0)
        finally:
            ops.ARENA_STORAGES.discard(o2.untyped_storage().data_ptr())
        assert torch.equal(o2, o1)
    finally:
        ops.set_gemm_mode(prev)


def test_a_split_k_accumulate_into_a_plain_buffer_is_not_deferred():
    """Inside ops.deferred_sums() the `C += partials` fold of a split-K launch is queued until the context's exit -- right for the
    optimizer's gradient arenas (nothing reads them before the step), wrong for any other destination, which the NEXT launch reads
    (how the shelved residual hand-over first failed the G4 golden; advisor, round 5). ops.gemm defers only arena destinations: a plain
    buffer takes the fold-free plan and is complete when the call returns to the stream."""
    M, N, K = 96, 64, 65536
    assert ops.gemm_plan(M, N, K, False, False)[1] > 1                     # the shape's own plan is split-K
    g = torch.Generator(device=DEV).manual_seed(3)
    A = torch.randn(K, M, device=DEV, generator=g); B = torch.randn(K, N, device=DEV, generator=g)
    C0 = torch.randn(M, N, device=DEV, generator=g)
    want = C0.double() + A.double().t() @ B.double()
    outs = {}
    for name, defer in (("eager", False), ("deferred", True)):
        C = C0.clone()
        if defer:
            with ops.deferred_sums():
                ops.gemm(A, B, False, False, M, N, K, out=C, ldc=N, accumulate=True)
                outs[name] = C.clone()                                     # read INSIDE the context, as the next launch would
        else:
            ops.gemm(A, B, False, False, M, N, K, out=C, ldc=N, accumulate=True)
            outs[name] = C.clone()
    for name, got in outs.items():
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), name
    # ... while an arena destination inside the context still gets the deferred fold (the 14 -> 2 merge launches of a step depend on it)
    arena = torch.zeros(M * N, device=DEV)
    ops.ARENA_STORAGES.add(arena.untyped_storage().data_ptr())
    try:
        with ops.deferred_sums() as ctx:
            ops.gemm(A, B, False, False, M, N, K, out=arena.view(M, N), ldc=N, accumulate=True)
            if ctx.on:
                assert _lib.lib().advmil_pending_sums(ctx.stream) >= 1
        assert float((arena.view(M, N).double() - A.double().t() @ B.double()).abs().max()) <= 2e-5 * float(want.abs().max())
    finally:
        ops.ARENA_STORAGES.discard(arena.untyped_storage().data_ptr())
