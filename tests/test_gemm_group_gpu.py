"""Several small deep-K weight-gradient contractions in one launch (advmil_gemm_tn_group, csrc/gemm_f32.hip): bit-identical to the plain
64x64-tile launches with the same split counts, close to float64, accumulating and overwriting members side by side, both arithmetic modes,
with the merges of the partial tiles deferred or not."""
import pytest
import torch

from advmil_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _splits(M, N, K):
    w = ((M + 63) // 64) * ((N + 63) // 64)
    sp = min(K // 256, 64)
    if sp * w > 1024:
        sp = 1024 // w
    return max(sp, 1)


@pytest.mark.parametrize("mode", ["bf16x3", "exact"])
@pytest.mark.parametrize("K", [64, 512, 8192, 32768 + 96])
@pytest.mark.parametrize("defer", [False, True])
def test_group_launch_equals_the_plain_launches(mode, K, defer):
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        g = torch.Generator().manual_seed(K)
        shapes = [(256, 128), (128, 64), (64, 128), (68, 36)]
        ops_ = []
        for (M, N) in shapes:
            A = torch.randn(K, M, generator=g).to(DEV)
            B = torch.randn(K, N, generator=g).to(DEV)
            ops_.append((A, B))
        base = [torch.randn(M, N, generator=g).to(DEV) for (M, N) in shapes]
        acc = [True, True, False, True]
        want = []
        for (A, B), c0, a, (M, N) in zip(ops_, base, acc, shapes):
            out = c0.clone()
            ops.gemm(A, B, False, False, M, N, K, out=out, ldc=N, accumulate=a, tile=11, splits=_splits(M, N, K))
            want.append(out)
        got = [c.clone() for c in base]
        if defer:
            with ops.deferred_sums():
                ops.gemm_tn_group([(A, B, o, a) for (A, B), o, a in zip(ops_, got, acc)])
        else:
            ops.gemm_tn_group([(A, B, o, a) for (A, B), o, a in zip(ops_, got, acc)])
        torch.cuda.synchronize()
        for w, o, (A, B), c0, a in zip(want, got, ops_, base, acc):
            assert torch.equal(w, o)
            ref = A.double().t() @ B.double() + (c0.double() if a else 0)
            assert float((o.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    finally:
        ops.set_gemm_mode(old)


def test_group_launch_into_strided_arena_views():
    """The region network's use: destinations are views of the gradient arena with their own row pitch."""
    K = 4096
    arena = torch.zeros(256 * 128 + 128 * 64, device=DEV)
    A1, B1 = torch.randn(K, 256, device=DEV), torch.randn(K, 128, device=DEV)
    A2, B2 = torch.randn(K, 128, device=DEV), torch.randn(K, 64, device=DEV)
    o1, o2 = arena[:256 * 128].view(256, 128), arena[256 * 128:].view(128, 64)
    ops.gemm_tn_group([(A1, B1, o1, True), (A2, B2, o2, True)])
    torch.cuda.synchronize()
    for o, A, B in ((o1, A1, B1), (o2, A2, B2)):
        ref = A.double().t() @ B.double()
        assert float((o.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
