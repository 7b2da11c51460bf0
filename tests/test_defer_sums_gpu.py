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
        with ops.deferred_sums():
            ops.gemm(a, b, False, False, M, N, K, out=o2, ldc=N, accumulate=True)
            assert _lib.lib().advmil_pending_sums(ops._stream()) == 1
            assert torch.equal(o2, base)
        assert torch.equal(o1, o2)
        ref = base.double() + a.double().t() @ b.double()
        assert float((o1.double() - ref).abs().max()) < 2e-2
    finally:
        ops.set_gemm_mode(prev)


def test_a_reader_inside_the_deferral_flushes_first():
    """The C-ABI contract of `accumulate` under deferral (advmil_hip.h): an accumulating split-K fold into ANY buffer (here not an arena
    slot) is postponed until the flush, so a caller that wants to read the destination before the deferral ends calls
    advmil_flush_sums -- after which it holds exactly what the undeferred call gives, and the queue is empty."""
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
        with ops.deferred_sums():
            ops.gemm(a, b, False, False, M, N, K, out=o2, ldc=N, accumulate=True)
            assert L.advmil_pending_sums(s) == 1 and torch.equal(o2, base)
            _lib.check(L.advmil_flush_sums(s), "flush_sums")
            assert L.advmil_pending_sums(s) == 0
            doubled = o2 * 2.0                                   # a reader on the same stream, behind the flush
            assert torch.equal(o2, o1) and torch.equal(doubled, o1 * 2.0)
        assert torch.equal(o2, o1)
    finally:
        ops.set_gemm_mode(prev)
