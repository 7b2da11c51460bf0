"""GPU: the fused ESAT attention core (advmil_mha_fwd / advmil_mha_bwd, csrc/attn.hip) and the post-norm residual kernels
(advmil_add_dropout_ln_fwd / _bwd) against a float64 torch-CPU restatement of the same op -- the op the reference gets from
nn.TransformerEncoderLayer (model/backbone_utils.py:113-127). Dropout masks are regenerated on the host from the kernels'
counter RNG (synth.attn_dropout_keep / synth.dropout_keep). Both arithmetic modes of the library are exercised; the attention
core itself always computes in split-bf16 (bf16x3), so its tolerance is that arithmetic's: relative 1e-5 forward, 5e-5 backward
(measured ~2e-6 / ~1e-5), far inside the path's 1e-4 contract."""
import numpy as np
import pytest
import torch

from advmil_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NH, HD, D = 8, 48, 384


@pytest.fixture(scope="module")
def ops():
    from advmil_amd import ops as _ops
    from advmil_amd import _lib
    _lib.lib()
    return _ops


@pytest.fixture(params=["exact", "bf16x3"])
def mode(request, ops):
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(prev)


@pytest.fixture(params=["two", "one"])
def bwd_form(request, ops):
    """Both backward forms of the attention core: advmil_mha_bwd (two launches, scores recomputed for dQ) and advmil_mha_bwd1
    (single pass, dQ through per-key-block partial slabs)."""
    prev = ops.ATTN_BWD
    ops.ATTN_BWD = request.param
    yield request.param
    ops.ATTN_BWD = prev


def rnd(tag, *shape, scale=1.0):
    n = int(np.prod(shape))
    return H.T(synth.normal(synth.stream_key(23, tag), n).reshape(shape) * np.float32(scale))


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def ref_attention(qkv, lens, masks=None, HD=HD):
    """float64: per bag and head softmax(q k^T / sqrt(head_dim)) (* mask) v on the packed [L_total, 3*8*head_dim] rows."""
    outs, r0 = [], 0
    D = NH * HD
    for b, L in enumerate(lens):
        blk = qkv[r0:r0 + L]
        q, k, v = (t.reshape(L, NH, HD).transpose(0, 1) for t in blk.split(D, dim=1))
        pr = torch.softmax(q @ k.transpose(-1, -2) / HD ** 0.5, dim=-1)
        if masks is not None:
            pr = pr * masks[b]
        outs.append((pr @ v).transpose(0, 1).reshape(L, D))
        r0 += L
    return torch.cat(outs, dim=0)


def host_masks(seed, sid, lens, p, rowoff=None):
    out, r0 = [], 0
    for b, L in enumerate(lens):
        rows = np.arange(L) + r0 + (0 if rowoff is None else int(rowoff[b]))
        m = np.stack([synth.attn_dropout_keep(seed, sid, rows, NH, h, L, p) for h in range(NH)])
        out.append(torch.from_numpy(m.astype(np.float64) / (1.0 - p)))
        r0 += L
    return out


@pytest.mark.parametrize("lens,p", [([32], 0.0), ([70], 0.0), ([210], 0.25), ([512], 0.25), ([128, 64, 200], 0.25),
                                    ([1, 5, 129], 0.0), ([2048], 0.25), ([700, 257, 33, 1025], 0.25)])
def test_mha_fused_fwd_bwd_vs_float64(ops, mode, bwd_form, lens, p):
    Lt = sum(lens)
    qkv = rnd(f"q{lens}", Lt, 3 * D, scale=0.7); go = rnd(f"g{lens}", Lt, D)
    seg = ops.Segments(lens, DEV) if len(lens) > 1 else None
    rng = ops.DeviceRng(DEV, seed=77)
    rng.record = True
    a = qkv.clone().to(DEV).requires_grad_(True)
    o = ops.mha(a, NH, p, rng, seg=seg)
    (o * go.to(DEV)).sum().backward()
    masks = None
    if p > 0:
        (_, sid, _, _), = [e for e in rng.log if e[0] == "mha_attn"]
        masks = host_masks(77, sid, lens, p)
    r = qkv.clone().double().requires_grad_(True)
    orf = ref_attention(r, lens, masks)
    (orf * go.double()).sum().backward()
    assert torch.isfinite(o).all() and torch.isfinite(a.grad).all()
    assert relerr(o, orf) < 1e-5, relerr(o, orf)
    assert relerr(a.grad, r.grad) < 5e-5, relerr(a.grad, r.grad)
    # per-block relative error too (q | k | v gradients have different magnitudes)
    for c in range(3):
        assert relerr(a.grad[:, c * D:(c + 1) * D], r.grad[:, c * D:(c + 1) * D]) < 5e-5, c


@pytest.mark.parametrize("hd", [16, 32, 64])
def test_mha_other_head_dims(ops, bwd_form, hd):
    """nn.TransformerEncoderLayer(d_model = bcb_dims[1], nhead = 8) for the other backbone widths the reference's load_backbone
    accepts (model/backbone.py:30-33): d_model 128 / 256 / 512 -> head_dim 16 / 32 / 64, ragged bags, dropout on."""
    lens, p, d = [130, 64, 300], 0.25, NH * hd
    Lt = sum(lens)
    qkv = rnd(f"hd{hd}", Lt, 3 * d, scale=0.7); go = rnd(f"hdg{hd}", Lt, d)
    seg = ops.Segments(lens, DEV)
    rng = ops.DeviceRng(DEV, seed=78)
    rng.record = True
    a = qkv.clone().to(DEV).requires_grad_(True)
    o = ops.mha(a, NH, p, rng, seg=seg)
    (o * go.to(DEV)).sum().backward()
    (_, sid, _, _), = [e for e in rng.log if e[0] == "mha_attn"]
    masks = host_masks(78, sid, lens, p)
    r = qkv.clone().double().requires_grad_(True)
    orf = ref_attention(r, lens, masks, HD=hd)
    (orf * go.double()).sum().backward()
    assert relerr(o, orf) < 1e-5, relerr(o, orf)
    for c in range(3):
        assert relerr(a.grad[:, c * d:(c + 1) * d], r.grad[:, c * d:(c + 1) * d]) < 5e-5, c


@pytest.mark.parametrize("lens,p", [([300], 0.0), ([700, 257, 33, 1025, 1], 0.25), ([2048, 2048], 0.25)])
def test_single_pass_backward_equals_the_two_launch_backward(ops, lens, p):
    """advmil_mha_bwd1 against advmil_mha_bwd on the same saved forward: dK / dV come out of the same arithmetic in a different
    order of 32-query steps, dQ is summed per 256-key block first -- fp32 round-off apart (1e-5 of each block's scale), every element."""
    Lt = sum(lens)
    qkv = rnd(f"sp{lens}", Lt, 3 * D, scale=0.7).to(DEV); go = rnd(f"spg{lens}", Lt, D).to(DEV)
    seg = ops.Segments(lens, DEV)
    grads = {}
    prev = ops.ATTN_BWD
    try:
        for form in ("two", "one"):
            ops.ATTN_BWD = form
            rng = ops.DeviceRng(DEV, seed=79)
            a = qkv.clone().requires_grad_(True)
            o = ops.mha(a, NH, p, rng, seg=seg)
            (o * go).sum().backward()
            grads[form] = a.grad.clone()
    finally:
        ops.ATTN_BWD = prev
    assert torch.isfinite(grads["one"]).all()
    for c in range(3):
        g1, g2 = grads["one"][:, c * D:(c + 1) * D], grads["two"][:, c * D:(c + 1) * D]
        assert float((g1 - g2).abs().max() / g2.abs().max()) < 1e-5, c


def test_attention_dropout_statistics():
    """The per-(query, 4 keys) byte hash of the attention dropout: keep rate, and no visible correlation along keys / queries."""
    m = np.stack([synth.attn_dropout_keep(5, 9, np.arange(1024), NH, h, 2048, 0.25) for h in range(2)]).astype(np.float64)
    assert abs(m.mean() - 0.75) < 1e-3
    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
    for lag in (1, 2, 3, 4, 8, 64):
        assert abs(corr(m[:, :, :-lag], m[:, :, lag:])) < 3e-3, lag
    assert abs(corr(m[:, :-1], m[:, 1:])) < 3e-3 and abs(corr(m[0], m[1])) < 3e-3
    assert abs(synth.attn_dropout_scale(0.25) - 1 / 0.75) < 1e-12


def test_mha_forced_rescale_and_large_scores(ops):
    """Online-softmax rescaling: one key per query block dominates late in the key order, so the running max jumps at a late
    tile (the rare branch a bounded random input never takes); plus scores of magnitude ~60."""
    L = 384
    qkv = rnd("spike", L, 3 * D, scale=0.3)
    q, k = qkv[:, :D], qkv[:, D:2 * D]
    k[300] = 6.0 * q[17]          # query 17 (and its neighbours in direction) meets a huge score at key 300 (5th key tile)
    k[70] = -4.0 * q[200]
    go = rnd("spikeg", L, D)
    a = qkv.clone().to(DEV).requires_grad_(True)
    o = ops.mha(a, NH, 0.0, None)
    (o * go.to(DEV)).sum().backward()
    r = qkv.clone().double().requires_grad_(True)
    orf = ref_attention(r, [L])
    (orf * go.double()).sum().backward()
    assert relerr(o, orf) < 1e-5 and relerr(a.grad, r.grad) < 5e-5


def test_mha_slab_equals_per_bag_and_rowoff_replays_global_rows(ops):
    """A ragged slab in one launch equals per-bag launches bit for bit without dropout; with dropout, a bag launched alone with
    `rowoff` = its global row offset draws exactly the mask it gets inside the slab (the bag-parallel invariance hook)."""
    lens = [96, 160, 40]
    Lt = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(Lt, 3 * D, device="cuda", generator=g)
    w = torch.randn(Lt, D, device="cuda", generator=g)
    seg = ops.Segments(lens, DEV)
    a = qkv.clone().requires_grad_(True)
    oa = ops.mha(a, NH, 0.0, None, seg=seg)
    (oa * w).sum().backward()
    b = qkv.clone().requires_grad_(True)
    offs = [0, 96, 256, 296]
    ob = torch.cat([ops.mha(b[offs[i]:offs[i + 1]], NH, 0.0, None) for i in range(3)], dim=0)
    (ob * w).sum().backward()
    assert torch.equal(oa, ob) and torch.equal(a.grad, b.grad)
    rng = ops.DeviceRng(DEV, seed=5)
    c = qkv.clone().requires_grad_(True)
    oc = ops.mha(c, NH, 0.25, rng, seg=seg)
    (oc * w).sum().backward()
    rng2 = ops.DeviceRng(DEV, seed=5)          # same seed, same call-site counter -> same stream id
    d = qkv[offs[1]:offs[2]].clone().requires_grad_(True)
    od = ops.mha(d, NH, 0.25, rng2, seg=ops.Segments([lens[1]], DEV), rowoff=torch.tensor([offs[1]], dtype=torch.int64, device=DEV))
    (od * w[offs[1]:offs[2]]).sum().backward()
    assert torch.equal(od, oc[offs[1]:offs[2]]) and torch.equal(d.grad, c.grad[offs[1]:offs[2]])


def test_mha_rejects_bad_args(ops):
    from advmil_amd._lib import AdvmilHipError
    with pytest.raises(AdvmilHipError):
        ops.mha(torch.randn(64, 3 * 320, device=DEV), 8)          # head_dim 40: only 16 / 32 / 48 / 64 are built
    with pytest.raises(RuntimeError):
        ops.mha(torch.randn(64, 3 * D), 8)                         # CPU tensor: no fallback


@pytest.mark.parametrize("R,d,p", [(64, 384, 0.0), (210, 384, 0.25), (37, 128, 0.25), (2048, 384, 0.25)])
def test_add_dropout_layer_norm_vs_float64(ops, R, d, p):
    x = rnd(f"lx{R}", R, d); o = rnd(f"lo{R}", R, d); gy = rnd(f"lg{R}", R, d)
    gamma = 1.0 + 0.1 * rnd("lgam", d); beta = 0.05 * rnd("lbet", d)
    rng = ops.DeviceRng(DEV, seed=9)
    rng.record = True
    xa, oa = x.clone().to(DEV).requires_grad_(True), o.clone().to(DEV).requires_grad_(True)
    ga, ba = gamma.clone().to(DEV).requires_grad_(True), beta.clone().to(DEV).requires_grad_(True)
    y = ops.add_dropout_layer_norm(xa, oa, ga, ba, 1e-5, p, rng, "ln_site")
    (y * gy.to(DEV)).sum().backward()
    keep = torch.ones(R, d, dtype=torch.float64)
    if p > 0:
        (_, sid, _, _), = [e for e in rng.log if e[0] == "ln_site"]
        keep = torch.from_numpy(synth.dropout_keep(9, sid, R * d, p).reshape(R, d).astype(np.float64)) / (1 - p)
    xr, orr = x.double().requires_grad_(True), o.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr + orr * keep, (d,), gr, br, 1e-5)
    (yr * gy.double()).sum().backward()
    assert relerr(y, yr) < 2e-6
    for got, want in ((xa.grad, xr.grad), (oa.grad, orr.grad), (ga.grad, gr.grad), (ba.grad, br.grad)):
        assert relerr(got, want) < 1e-5, relerr(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [False, True])
def test_in_projection_writes_qkv_as_operand_planes_only(ops, training):
    """The ESAT layer's in-projection feeds the attention kernels and nothing else, and they read q | k | v as bf16x3 operand planes: in
    bf16x3 mode the contraction's epilogue writes the planes INSTEAD of the fp32 values (advmil_gemm_f32_tiled with C == NULL) and the
    split pass in front of the attention launch is gone. The planes are the split of the very values the fp32 form would have stored, so
    the layer's output and every gradient are EQUAL with the switch on and off (shipped dropout on in the training case: same draws)."""
    from advmil_amd.model.esat import HipTransformerEncoderLayer
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        torch.manual_seed(5)
        layer = HipTransformerEncoderLayer(384, 8, 384, 0.25).cuda()
        layer.train(training)
        g = torch.Generator(device="cuda").manual_seed(9)
        lens = [2048, 1536, 512, 1040]                    # 5136 rows: slab-sized (>= 4096), ragged
        x = torch.randn(sum(lens), 384, device="cuda", generator=g)
        go = torch.randn(sum(lens), 384, device="cuda", generator=g)
        seg = ops.Segments(lens, x.device)
        outs = []
        for on in (False, True):
            ops.ATTN_QKV_PLANES = on
            layer.rng = ops.DeviceRng(x.device, seed=77)
            for p_ in layer.parameters():
                p_.grad = None
            xi = x.clone().requires_grad_(True)
            seen = []
            real = ops.split_planes
            ops.split_planes = lambda t, *a, **k: (seen.append(tuple(t.shape)), real(t, *a, **k))[1]
            try:
                y = layer.forward_rows(xi, seg)
                (y * go).sum().backward()
            finally:
                ops.split_planes = real
            assert ((sum(lens), 1152) in seen) == (not on)          # the split pass of qkv runs only with the switch off
            outs.append((y.detach().clone(), xi.grad.clone(), [p_.grad.clone() for p_ in layer.parameters()]))
        (y0, gx0, gp0), (y1, gx1, gp1) = outs
        assert torch.isfinite(y1).all() and torch.equal(y0, y1) and torch.equal(gx0, gx1)
        assert all(torch.equal(a, b) for a, b in zip(gp0, gp1))
    finally:
        ops.ATTN_QKV_PLANES = True
        ops.set_gemm_mode(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("hd,lens", [(48, [700, 257, 33, 1025, 1]), (32, [130, 64, 300]), (64, [2048]), (16, [5, 129])])
def test_forward_with_the_log_sum_exp_given_equals_the_plain_forward(ops, hd, lens):
    """advmil_mha_fwd_lse: the train-mode forward that takes the softmax statistics from the eval-mode pass over the same q | k | v
    (the handler's forward memo) -- probabilities exp2(s c - lse) instead of the running maximum / rescale / row sum. Same dropout
    stream: output within round-off of the plain train-mode forward -- the given statistic is one fp32 number per (query, head), its
    rounding moves every probability of the row by ~|lse| 2^-24 -- asserted at 1e-5 of the scale (the float64 tests' own forward bound),
    and the backward it feeds likewise."""
    d = NH * hd
    Lt = sum(lens)
    qkv = rnd(f"lse{hd}{lens}", Lt, 3 * d, scale=0.7).to(DEV); go = rnd(f"lseg{hd}{lens}", Lt, d).to(DEV)
    seg = ops.Segments(lens, DEV)
    planes = ops.split_planes(qkv)
    with torch.no_grad():
        ops.MhaFn.apply(qkv, NH, 0.0, None, 0, seg, None, planes)                     # eval-mode pass: leaves the log-sum-exp
    lse = ops.MhaFn.last_lse
    res = []
    for given in (None, lse):
        rng = ops.DeviceRng(DEV, seed=81)
        sid = rng.site("mha_attn", (Lt, NH), 0.25)
        a = qkv.clone().requires_grad_(True)
        o = ops.MhaFn.apply(a, NH, 0.25, rng.seed, sid, seg, None, planes, given)
        assert (ops.MhaFn.last_lse is lse) == (given is not None)
        (o * go).sum().backward()
        res.append((o.detach().clone(), a.grad.clone()))
    (o0, g0), (o1, g1) = res
    assert torch.isfinite(o1).all() and torch.isfinite(g1).all()
    assert float((o0 - o1).abs().max()) <= 1e-5 * float(o0.abs().max())
    for c in range(3):
        a_, b_ = g0[:, c * d:(c + 1) * d], g1[:, c * d:(c + 1) * d]
        assert float((a_ - b_).abs().max()) <= 1e-5 * float(a_.abs().max())
