"""Round 6: the generator's first layer over the step slab leaves its [rows, hid] output as bf16x3 operand planes ONLY (ops.H_PLANES_ONLY):
the pooling kernels and the dropout replay of the memoized output read / write planes (model/backbone.py:60-66, 79-86 run twice per
optimizer step: model_handler.py:398-400 eval under no_grad, 420-425 train). Kernel level: equal (to an ulp) to the fp32-row kernels fed hi + lo.
Step level: the gradients of a G + D step agree with the fp32-row path to the 2^-17 of the bf16x3 arithmetic."""
import numpy as np
import pytest
import torch

from advmil_amd import ops
from tests import helpers as H
from tests.test_parity_gpu import DEV, make_handler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens,D", [([8192] * 4, 384), ([4096, 1008, 16, 7312, 2000], 384), ([640, 4096], 128)])
def test_pooling_from_planes_equals_pooling_of_hi_plus_lo(lens, D):
    N = sum(lens)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(N, D, generator=g).relu_().to(DEV)
    s = (torch.randn(N, generator=g) * 3).to(DEV)
    pl = ops.split_planes(h)
    hr = ops.planes_f32(pl)                                   # what the plane-fed call sees: hi + lo (exact in fp32)
    assert float((hr - h).abs().max()) <= 2.0 ** -16 * float(h.abs().max())
    seg = ops.Segments(lens, DEV)
    A0, p0 = ops.softmax_pool(s, hr, N, D, seg)
    A1, p1 = ops.softmax_pool(s, hr, N, D, seg, pl)
    # (same values, same order of the rows; the two instantiations contract their multiply-adds differently: 1 ulp)
    assert torch.equal(A0, A1) and float((p0 - p1).abs().max()) <= 2.5e-7 * float(p0.abs().max())
    ref = torch.stack([(torch.softmax(s[a:b].double(), 0)[None, :] @ hr[a:b].double())[0] for a, b in zip(np.cumsum([0] + lens[:-1]), np.cumsum(lens))])
    assert float((p1.double() - ref).abs().max()) < 2e-6
    dp = torch.randn(len(lens), D, generator=g).to(DEV)
    dA = torch.randn(N, generator=g).to(DEV)
    d0 = ops.softmax_pool_bwd(dp, dA, A0, hr, N, D, seg)
    d1 = ops.softmax_pool_bwd(dp, dA, A0, hr, N, D, seg, pl)
    assert float((d0 - d1).abs().max()) <= 1e-6 * float(d0.abs().max())


@pytest.mark.parametrize("M,N", [(4096, 384), (1000, 128), (131072, 384)])
def test_dropout_of_planes_equals_the_fp32_replay(M, N):
    """advmil_dropout_planes against advmil_act_dropout_bwd's replay (the round-5 path) on the same values and the same draw."""
    g = torch.Generator().manual_seed(8)
    y0 = ops.planes_f32(ops.split_planes(torch.randn(M, N, generator=g).relu_().to(DEV)))      # values that ARE hi + lo
    pl = ops.split_planes(y0)
    rng = ops.DeviceRng(DEV, seed=21)
    rr = torch.randperm(M, generator=g).to(DEV) if M == 1000 else None
    tpl, tbits = ops.dropout_planes(pl, M, N, 0.25, rng.seed, 5, rr)
    rpl = ops.Planes.alloc((M, N), DEV)
    rbits = torch.empty(M, N // 32, dtype=torch.int32, device=DEV)
    yr, _ = ops.act_dropout_bwd(y0, y0, ops.ACT_NONE, M, N, 0.25, rng.seed, 5, want_bias=False, planes=rpl, bits=rbits, rng_row=rr)
    torch.cuda.synchronize()
    assert torch.equal(tpl.hi, rpl.hi) and torch.equal(tpl.lo, rpl.lo) and torch.equal(tbits, rbits)
    assert torch.equal(ops.planes_f32(tpl) > 0, yr > 0)
    assert 0.30 < float((tpl.hi != 0).float().mean()) < 0.45


def _step(planes_only, nb=8, n=8192, drop=True):
    old = ops.H_PLANES_ONLY
    ops.H_PLANES_ONLY = planes_only
    prev = ops.get_gemm_mode()
    try:
        h, _, _ = make_handler("abmil", bp_every_batch=nb, gemm_mode="bf16x3")
        if not drop:
            from tests.test_parity_gpu import zero_dropout
            zero_dropout(h.netG); zero_dropout(h.netD)
        xs = [[H.bag(i, n, DEV), torch.zeros(1, 1, device=DEV)] for i in range(nb)]
        ys_host = [H.label(i) for i in range(nb)]
        ys = [y.to(DEV) for y in ys_host]
        h.rng.reset(5)
        plan = h._plan(xs, ys, "wlabel", None, ys_host)
        seen = []
        real = ops.softmax_pool

        def spy(s, hh, N, D, seg=None, hpl=None):
            seen.append((N, D, hpl is not None))
            return real(s, hh, N, D, seg, hpl)
        ops.softmax_pool = spy
        try:
            preds, fakes = h._disc_backward(0, xs, ys, plan)
            gd = h.optimizerD.flat_grad.clone()
            h.optimizerD.step()
            h._gen_backward(0, xs, ys, plan)
        finally:
            ops.softmax_pool = real
        torch.cuda.synchronize()
        return (torch.cat([q.detach().reshape(-1) for q in preds]).clone(), gd, {k: p.grad.clone() for k, p in h.netG.named_parameters()}, seen,
                h.netG.backbone.last_attention.clone())
    finally:
        ops.H_PLANES_ONLY = old
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("drop", [True, False])
def test_step_with_planes_only_h_matches_the_fp32_row_step(drop):
    pa, da, ga, sa, Aa = _step(True, drop=drop)
    pb, db, gb, sb, Ab = _step(False, drop=drop)
    big = [t for t in sa if t[0] == 8 * 8192 and t[1] == 384]
    assert len(big) == 2 and all(t[2] for t in big), sa              # both generator passes pooled from planes ...
    assert not any(t[2] for t in sb), sb                             # ... and neither with the switch off
    assert float((pa - pb).abs().max()) < 2e-6
    assert float((Aa - Ab).abs().max()) < 1e-7
    assert float((da - db).abs().max()) <= 2e-5 * float(db.abs().max()) + 1e-9
    for k in ga:
        a, b = ga[k].double(), gb[k].double()
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-8, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("kind", ["patch", "abmil"])
def test_chained_forward_memo_is_bitwise_neutral(kind):
    """ForwardMemo.derived (round 6): behind the generator's first layer the record pass's LayerNorm / ReLU / mean16 and the ESAT
    in-projection are carried into the train-mode forward too (everything in front of the transformer layer's first dropout;
    model/backbone_utils.py:158-168, 113-127). Skipping a deterministic recomputation must not move a bit: two optimizer steps with
    the chain == two steps without, dropout ON."""
    from advmil_amd.config import default_cfg
    from advmil_amd.model import MyHandler
    from tests.test_parity_gpu import load_synth

    def run(chain):
        old = ops.MEMO_CHAIN
        ops.MEMO_CHAIN = chain
        try:
            nb, n = 4, 8192
            calls = []
            real = ops.ln_relu_mean16_fwd
            ops.ln_relu_mean16_fwd = lambda *a, **k: (calls.append(a[3]), real(*a, **k))[1]
            h = MyHandler(default_cfg(bcb_mode=kind, bp_every_batch=nb, gemm_mode="bf16x3"), device=DEV)
            load_synth(h.netG, f"G-{kind}:"); load_synth(h.netD, "D-prj:")
            h.optimizerG.refresh_planes(); h.optimizerD.refresh_planes()
            h.rng.reset(99)
            xs = [[H.bag(i, n, DEV), torch.zeros(1, 1, device=DEV)] for i in range(nb)]
            ys_host = [H.label(i) for i in range(nb)]
            ys = [y.to(DEV) for y in ys_host]
            for i in range(2):
                plan = h._plan(xs, ys, "wlabel", None, ys_host)
                h._update_disc(i, xs, ys, "wlabel", None, ys_host=ys_host, plan=plan)
                h._update_gen(i, xs, ys, "wlabel", None, ys_host=ys_host, plan=plan)
                h.rng.advance(1)
            torch.cuda.synchronize()
            return (h.pop_logs(), h.optimizerG.flat_param.clone(), h.optimizerD.flat_param.clone(), len(calls))
        finally:
            ops.MEMO_CHAIN = old
            ops.ln_relu_mean16_fwd = real
            ops.set_gemm_mode("exact")

    a, b = run(True), run(False)
    for la, lb in zip(a[0], b[0]):
        for k in lb:
            assert float(la[k]) == float(lb[k]), k
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    if kind == "patch":
        assert a[3] == b[3] - 2, (a[3], b[3])           # one LayerNorm/ReLU/mean16 launch less per optimizer step


def _gate_step(fused, nb=8, n=8192):
    old = ops.FUSED_GATE_TRAIN
    ops.FUSED_GATE_TRAIN = fused
    prev = ops.get_gemm_mode()
    try:
        h, _, _ = make_handler("abmil", bp_every_batch=nb, gemm_mode="bf16x3")
        xs = [[H.bag(i, n, DEV), torch.zeros(1, 1, device=DEV)] for i in range(nb)]
        ys_host = [H.label(i) for i in range(nb)]
        ys = [y.to(DEV) for y in ys_host]
        h.rng.reset(5)
        plan = h._plan(xs, ys, "wlabel", None, ys_host)
        seen = []
        real = ops.gate_score

        def spy(*a, **k):
            seen.append(a[3:5])
            return real(*a, **k)
        ops.gate_score = spy
        try:
            preds, _ = h._disc_backward(0, xs, ys, plan)
            h.optimizerD.step()
            h._gen_backward(0, xs, ys, plan)
        finally:
            ops.gate_score = real
        torch.cuda.synchronize()
        return (torch.cat([q.detach().reshape(-1) for q in preds]).clone(), {k: p.grad.clone() for k, p in h.netG.named_parameters()}, seen,
                h.netG.backbone.last_attention.clone())
    finally:
        ops.FUSED_GATE_TRAIN = old
        ops.set_gemm_mode(prev)


def test_training_gate_score_in_the_contraction_epilogue_equals_the_score_pass():
    """Round 6 (ops.FUSED_GATE_TRAIN): the training pass of Attn_Net_Gated (model/backbone_utils.py:24-29, dropout 0.25 on both branches)
    stores its activations in pair blocks of 32 columns and reduces the score in the contraction's epilogue from keep bits drawn by the
    first layer's dropout launch; the backward reads the blocked layout and the weight gradient's merge un-permutes its rows. Same
    draws, same products: attention weights and every generator gradient agree with the score-pass form to fp32 round-off."""
    pa, ga, sa, Aa = _gate_step(True)
    pb, gb, sb, Ab = _gate_step(False)
    slab = 8 * 8192
    assert not any(t[0] == slab for t in sa), sa                 # no pass over the stored [rows, 2D] activations ...
    assert any(t[0] == slab for t in sb), sb                     # ... which the unfused form makes
    assert float((pa - pb).abs().max()) < 1e-6
    assert float((Aa - Ab).abs().max()) < 2e-7 and float(Aa.sum()) > 0
    for k in ga:
        a, b = ga[k].double(), gb[k].double()
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-8, (k, float((a - b).abs().max()), scale)
