"""GPU: large-size, EVERY-ELEMENT checks of the kernels that evaluate hardware transcendentals (v_exp_f32 / v_rcp_f32 / v_rsq_f32 /
v_log_f32). Found in round 2: in the fused gate-score epilogue of the bf16x3 contraction kernels ~0.1 % of the wavefronts consumed
a transcendental's result before its last 16-lane pass (lanes 48-63) had landed -- rows 6 and 7 of every 32-row sub-tile were off
by 1-15 % in ~250 of 10^6 entries at 131072 rows, never at the <= 8192-row sizes the earlier tests used (and invisible to strided
comparisons). csrc/common.h::hw_* now pads every such op; these tests keep the failure mode covered: full comparisons at the
bench's slab size, several launches each (the error was timing dependent)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from advmil_amd import ops as _ops
    from advmil_amd import _lib
    _lib.lib()
    return _ops


@pytest.mark.parametrize("mode", ["bf16x3", "exact"])
def test_fused_gate_score_every_entry_at_slab_size(ops, mode):
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        M, D = 131072, 384
        g = torch.Generator(device=DEV).manual_seed(0)
        h = torch.randn(M, D, device=DEV, generator=g)
        Wi = torch.randn(2 * D, D, device=DEV, generator=g) * 0.05
        bi = torch.randn(2 * D, device=DEV, generator=g) * 0.1
        wc = torch.randn(D, device=DEV, generator=g)
        C = ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi).double()
        ref = (torch.tanh(C[:, 0::2]) * torch.sigmoid(C[:, 1::2]) * wc[None, :].double()).sum(1)
        tiles = [0, 23, 22, 13, 12, 11] + ([43, 42] if mode == "bf16x3" else [])
        for tile in tiles:
            for _ in range(3):
                s = ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc, tile=tile).double().sum(1)
                d = (s - ref).abs()
                assert int((d > 2e-4).sum()) == 0, (tile, int((d > 2e-4).sum()), float(d.max()))
        if mode == "bf16x3":                                   # the plane-fed LDS-DMA kernel's gate mode
            ph, pw = ops.split_planes(h), ops.split_planes(Wi)
            for tile in (0, 82, 83, 84):
                for _ in range(3):
                    s = ops.gemm(h, Wi, True, True, M, 2 * D, D, bias=bi, gate_wc=wc, a_planes=ph, b_planes=pw, tile=tile,
                                 splits=1 if tile else None).double().sum(1)
                    d = (s - ref).abs()
                    assert int((d > 2e-4).sum()) == 0, (tile, int((d > 2e-4).sum()), float(d.max()))
    finally:
        ops.set_gemm_mode(prev)


def test_gemm_activation_epilogue_every_entry_at_slab_size(ops):
    """tanh | sigmoid epilogue (the training pass of the gate branches) on the full slab, all entries, both arithmetic modes."""
    M, D = 131072, 384
    g = torch.Generator(device=DEV).manual_seed(1)
    h = torch.randn(M, D, device=DEV, generator=g)
    W = torch.randn(2 * D, D, device=DEV, generator=g) * 0.05
    b = torch.randn(2 * D, device=DEV, generator=g) * 0.1
    prev = ops.get_gemm_mode()
    try:
        for mode in ("bf16x3", "exact"):
            ops.set_gemm_mode(mode)
            pre = ops.gemm(h, W, True, True, M, 2 * D, D, bias=b).double()
            ref = torch.cat([torch.tanh(pre[:, :D]), torch.sigmoid(pre[:, D:])], dim=1)
            for _ in range(3):
                ab = ops.gemm(h, W, True, True, M, 2 * D, D, bias=b, act0=2, act1=3, act_split=D).double()
                assert float((ab - ref).abs().max()) < 5e-6
    finally:
        ops.set_gemm_mode(prev)


def test_softmax_pool_and_layernorm_every_entry_at_slab_size(ops):
    rows, D, bags = 131072, 384, 16
    g = torch.Generator(device=DEV).manual_seed(2)
    hh = torch.randn(rows, D, device=DEV, generator=g)
    s = torch.randn(rows, device=DEV, generator=g) * 3.0
    seg = ops.Segments([rows // bags] * bags, DEV)
    ref = torch.softmax(s.double().reshape(bags, -1), dim=1).reshape(-1)
    for _ in range(3):
        A, pooled = ops.softmax_pool(s, hh, rows, D, seg)
        assert float(((A.double() - ref).abs() / ref).max()) < 2e-5          # relative, every attention weight
        want = (ref.reshape(bags, -1, 1) * hh.double().reshape(bags, -1, D)).sum(1)
        assert float((pooled.double() - want).abs().max()) < 1e-5
    x = torch.randn(32768, D, device=DEV, generator=g)
    o = torch.randn(32768, D, device=DEV, generator=g)
    gm = 1.0 + 0.1 * torch.randn(D, device=DEV, generator=g); bt = 0.05 * torch.randn(D, device=DEV, generator=g)
    want = torch.nn.functional.layer_norm((x + o).double(), (D,), gm.double(), bt.double(), 1e-5)
    for _ in range(3):
        y = ops.add_dropout_layer_norm(x, o, gm, bt, 1e-5, 0.0)
        assert float((y.double() - want).abs().max()) < 5e-6
    yv = torch.randn(rows, 128, device=DEV, generator=g)
    g1 = 1.0 + 0.1 * torch.randn(128, device=DEV, generator=g); b1 = 0.05 * torch.randn(128, device=DEV, generator=g)
    want = torch.relu(torch.nn.functional.layer_norm(yv.double(), (128,), g1.double(), b1.double(), 1e-5)).reshape(-1, 16, 128).mean(1)
    for _ in range(3):
        emb = ops.ln_relu_mean16(yv, g1, b1)
        assert float((emb.double() - want).abs().max()) < 5e-6


def test_attention_every_entry_at_configs3_size(ops):
    """The fused attention core at L = 2048 x 4 bags against float64 torch ON THE GPU (all elements, forward and backward)."""
    L, G, D, NH = 2048, 4, 384, 8
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(G * L, 3 * D, device=DEV, generator=g) * 0.7
    go = torch.randn(G * L, D, device=DEV, generator=g)
    seg = ops.Segments([L] * G, DEV)
    r = qkv.double().requires_grad_(True)
    outs = []
    for b in range(G):
        q, k, v = (t.reshape(L, NH, 48).transpose(0, 1) for t in r[b * L:(b + 1) * L].split(D, dim=1))
        outs.append((torch.softmax(q @ k.transpose(-1, -2) / 48 ** 0.5, dim=-1) @ v).transpose(0, 1).reshape(L, D))
    orf = torch.cat(outs)
    (orf * go.double()).sum().backward()
    for _ in range(2):
        a = qkv.clone().requires_grad_(True)
        o = ops.mha(a, NH, 0.0, None, seg=seg)
        (o * go).sum().backward()
        assert float((o.double() - orf).abs().max()) < 1e-5 * float(orf.abs().max())
        assert float((a.grad.double() - r.grad).abs().max()) < 5e-5 * float(r.grad.abs().max())


def test_plane_fed_kernel_race_screen(ops):
    """The LDS-DMA staged NT kernel (global_load_lds into two LDS buffers, one barrier per 32-deep chunk) against the register-staged
    generic kernel: bit-identical on every element, repeated launches at the slab shapes (a DMA / ds_read ordering slip would show as
    rare wrong tiles that come and go between runs)."""
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        g = torch.Generator(device=DEV).manual_seed(4)
        for M, N, K in ((131072, 384, 1024), (131072, 768, 384), (131072, 128, 1024), (32768, 1152, 384)):
            A = torch.randn(M, K, device=DEV, generator=g)
            B = torch.randn(N, K, device=DEV, generator=g)
            bias = torch.randn(N, device=DEV, generator=g)
            ref = ops.gemm(A, B, True, True, M, N, K, bias=bias, act0=1)
            pa, pb = ops.split_planes(A), ops.split_planes(B)
            tiles = [t for t in (82, 83) if N % (64 * (t - 80)) == 0]
            for t in tiles:
                for _ in range(4):
                    got = ops.gemm(A, B, True, True, M, N, K, bias=bias, act0=1, a_planes=pa, b_planes=pb, tile=t, splits=1)
                    assert torch.equal(got, ref), (M, N, K, t, int((got != ref).sum()))
    finally:
        ops.set_gemm_mode(prev)
