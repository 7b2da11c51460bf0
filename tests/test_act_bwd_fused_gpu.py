"""The first layer's activation / dropout backward in the epilogue of the gated-attention pool's dh contraction (advmil_epilogue_t: rowv +
maskref + colsum in one launch; ops.ACT_BWD_IN_DH): equal to the separate row pass it replaces (model/backbone.py:79-86 autograd)."""
import pytest
import torch

from advmil_amd import ops
from tests import helpers as H
from tests.test_parity_gpu import DEV, make_handler

pytestmark = pytest.mark.gpu


def _g_grads(fused, nb, n, kind="abmil"):
    old = ops.ACT_BWD_IN_DH
    ops.ACT_BWD_IN_DH = fused
    prev = ops.get_gemm_mode()
    try:
        h, _, _ = make_handler(kind, bp_every_batch=nb, gemm_mode="bf16x3")
        xs = [[H.bag(i, n, DEV), torch.zeros(1, 1, device=DEV)] for i in range(nb)]
        ys_host = [H.label(i) for i in range(nb)]
        ys = [y.to(DEV) for y in ys_host]
        h.rng.reset(5)
        plan = h._plan(xs, ys, "wlabel", None, ys_host)
        h._disc_backward(0, xs, ys, plan)
        h.optimizerD.step()
        launches = []
        real = ops.act_dropout_bwd

        def spy(*a, **k):
            launches.append(a[3:5])
            return real(*a, **k)
        ops.act_dropout_bwd = spy
        try:
            h._gen_backward(0, xs, ys, plan)
        finally:
            ops.act_dropout_bwd = real
        torch.cuda.synchronize()
        return h.optimizerG.flat_grad.clone(), {k: p.grad.clone() for k, p in h.netG.named_parameters()}, launches
    finally:
        ops.ACT_BWD_IN_DH = old
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("nb,n", [(8, 8192), (3, 4352)])
def test_first_layer_backward_in_the_dh_epilogue_equals_the_row_pass(nb, n):
    ga, pa, la = _g_grads(True, nb, n)
    gb, pb, lb = _g_grads(False, nb, n)
    slab = nb * n
    fused_here = slab % 256 == 0 and slab >= 65536        # (the slab's operand planes exist from one full wave of 256-row tiles on)
    # the slab-sized activation backward pass ((rows, 384) with the dropout replay's own call excluded: that one has act NONE) is gone
    big_a = [s for s in la if s[0] == slab]
    big_b = [s for s in lb if s[0] == slab]
    assert len(big_b) == len(big_a) + (1 if fused_here else 0), (la, lb)
    for k in pa:
        a, b = pa[k].double(), pb[k].double()
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (k, float((a - b).abs().max()), scale)
    assert float(ga.abs().max()) > 0


@pytest.mark.parametrize("M,N", [(4096, 384), (517, 128), (33, 1056)])
def test_dropout_pass_leaves_the_keep_mask_as_one_bit_per_element(M, N):
    g = torch.Generator().manual_seed(2)
    y0 = torch.relu(torch.randn(M, N, generator=g)).to(DEV)
    rng = ops.DeviceRng(DEV, seed=9)
    bits = torch.zeros(M, N // 32, dtype=torch.int32, device=DEV)
    y, _ = ops.act_dropout_bwd(y0, y0, ops.ACT_NONE, M, N, 0.25, rng.seed, 3, want_bias=False, bits=bits)
    torch.cuda.synchronize()
    b = bits.cpu().numpy().view("uint32")
    import numpy as np
    un = ((b[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(M, N).astype(bool)
    assert (un == (y.cpu().numpy() > 0)).all()
    assert 0.3 < un.mean() < 0.45            # relu of a normal (1/2) x kept (3/4)
