"""CPU: the negative paths of the C ABI (include/advmil_hip.h: "<0 = ADVMIL_E* (bad argument)"). Argument validation runs before
anything is enqueued, so these calls need no GPU: every launching entry point must answer ADVMIL_EINVAL (-1) to null pointers /
empty shapes, and the shape-, pitch-, alignment- and workspace-checks of each entry-point group must fire (a positive return
would be a hipError_t, i.e. the call got as far as a launch)."""
import ctypes

import pytest

EINVAL, EWORKSPACE = -1, -2
HOST_ONLY = {"advmil_version", "advmil_adam_blocks", "advmil_set_gemm_mode", "advmil_get_gemm_mode", "advmil_gemm_f32_plan", "advmil_gemm_f32_plan_layout",
             "advmil_gemm_f32_plan_planes", "advmil_gemm_f32_gate_blocks",
             # merge-queue bookkeeping on a stream handle (NULL = the default stream is a valid one): nothing to validate, nothing launched
             "advmil_defer_sums", "advmil_flush_sums", "advmil_pending_sums"}
A16 = 0x7F0000001000          # a fake, 16-byte aligned "device address": never dereferenced when validation does its job


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    g.build()
    from advmil_amd import _lib
    return _lib


def _zero_args(argtypes):
    out = []
    for t in argtypes:
        if t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            out.append(None)
        elif t is ctypes.c_float:
            out.append(0.0)
        else:
            out.append(0)
    return out


def test_every_launching_entry_point_rejects_null_pointers_and_empty_shapes(L):
    lib = L.lib()
    checked = 0
    for name, (res, argtypes) in L.SIGNATURES.items():
        if res is not ctypes.c_int or name in HOST_ONLY or name.endswith("_workspace_bytes"):
            continue
        rc = getattr(lib, name)(*_zero_args(argtypes))
        assert rc == EINVAL, (name, rc)
        checked += 1
    assert checked >= 30


def test_contraction_engine_checks_pitch_and_alignment(L):
    lib = L.lib()
    e = L.Epilogue()
    e.alpha = 1.0
    e.act_split = 1 << 30
    args = lambda A=A16, lda=64, B=A16 + (1 << 20), ldb=64, C=A16 + (2 << 20), ldc=64, M=64, N=64, K=64: (  # noqa: E731
        1, 1, M, N, K, ctypes.c_void_p(A), lda, ctypes.c_void_p(B), ldb, ctypes.c_void_p(C), ldc, ctypes.byref(e), 1, 0, None, 0, None)
    assert lib.advmil_gemm_f32_tiled(*args(lda=32)) == EINVAL                 # row pitch shorter than the row
    assert lib.advmil_gemm_f32_tiled(*args(ldc=48)) == EINVAL
    assert lib.advmil_gemm_f32_tiled(*args(A=A16 + 4)) == EINVAL              # operands must be 16-byte aligned
    assert lib.advmil_gemm_f32_tiled(*args(lda=66)) == EINVAL                 # ... and so must every row (pitch % 4)
    assert lib.advmil_gemm_f32_tiled(*args(M=0)) == EINVAL
    # split-K needs its workspace
    a = list(args(M=384, N=384, K=65536, lda=65536, ldb=65536, ldc=384))
    a[12] = 16
    need = lib.advmil_gemm_f32_workspace_bytes(384, 384, 16)
    assert need > 0
    a[14], a[15] = ctypes.c_void_p(A16 + (3 << 20)), need - 4
    assert lib.advmil_gemm_f32_tiled(*a) == EWORKSPACE
    assert lib.advmil_set_gemm_mode(7) == EINVAL
    t, s = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.advmil_gemm_f32_plan_layout(1, 1, 0, 64, 64, ctypes.byref(t), ctypes.byref(s)) == EINVAL


def test_pooling_group_checks_shapes_and_workspace(L):
    lib = L.lib()
    p = lambda k: ctypes.c_void_p(A16 + (k << 20))   # noqa: E731
    N, D = 8192, 384
    need = lib.advmil_softmax_pool_workspace_bytes(N, D, 1)
    assert lib.advmil_softmax_pool_fwd(p(0), p(1), D, N, D, 1, None, N, p(2), p(3), p(4), need - 4, None) == EWORKSPACE
    assert lib.advmil_softmax_pool_fwd(p(0), p(1), D - 4, N, D, 1, None, N, p(2), p(3), p(4), need, None) == EINVAL      # pitch < D
    assert lib.advmil_softmax_pool_fwd(p(0), p(1), D, N, D, 4, None, N, p(2), p(3), p(4), need, None) == EINVAL          # 4 bags, no offsets
    assert lib.advmil_softmax_pool_fwd(p(0), ctypes.c_void_p(A16 + 8), D, N, D, 1, None, N, p(2), p(3), p(4), need, None) == EINVAL
    assert lib.advmil_gate_score_fwd(p(0), p(1), p(2), 1.5, p(3), 1, 2, N, D, p(4), None, None) == EINVAL                # p >= 1
    assert lib.advmil_ln_relu_mean16_fwd(p(0), p(1), p(2), 1e-5, 8200, 128, p(3), p(4), p(5), None, None, 1, None, 0, None) == EINVAL   # N % 16 != 0


def test_attention_group_checks_head_dim_and_segments(L):
    lib = L.lib()
    p = lambda k: ctypes.c_void_p(A16 + (k << 20))   # noqa: E731
    fwd = lambda hd=48, Lt=512, nseg=1, mlen=512, pd=0.0, seed=None, hi=p(0): lib.advmil_mha_fwd(   # noqa: E731
        hi, p(1), Lt, 8, hd, nseg, None, mlen, pd, seed, 0, None, p(2), p(3), None)
    assert fwd(hd=40) == EINVAL and fwd(hd=96) == EINVAL                      # head_dim in {16, 32, 48, 64}
    assert fwd(nseg=4, mlen=128) == EINVAL                                    # 4 bags, no offsets
    assert fwd(mlen=1024) == EINVAL                                           # max_len > rows
    assert fwd(pd=1.0, seed=p(4)) == EINVAL                                   # p >= 1
    assert fwd(hi=ctypes.c_void_p(A16 + 8)) == EINVAL                         # planes are read in 16-byte units
    need = lib.advmil_mha_bwd_workspace_bytes(512, 8, 48)
    assert need >= 512 * 8 * 4 + 2 * 512 * 384 * 2
    assert lib.advmil_mha_bwd(p(0), p(1), p(2), p(3), p(4), 512, 8, 48, 1, None, 512, 0.0, None, 0, None, p(5), p(6), need - 16, None) == EWORKSPACE
    # the single-pass backward: its workspace also holds ceil(max_len / 256) partial slabs of dQ; same argument checks
    need1 = lib.advmil_mha_bwd1_workspace_bytes(512, 8, 48, 512)
    assert need1 >= need + 2 * 512 * 384 * 4
    assert lib.advmil_mha_bwd1_workspace_bytes(512, 8, 48, 100) == need + 1 * 512 * 384 * 4        # one 256-key block per bag
    bwd1 = lambda hd=48, nseg=1, mlen=512, ws=need1, out=p(2): lib.advmil_mha_bwd1(   # noqa: E731
        p(0), p(1), out, p(3), p(4), 512, 8, hd, nseg, None, mlen, 0.0, None, 0, None, p(5), p(6), ws, None)
    assert bwd1(ws=need1 - 16) == EWORKSPACE
    assert bwd1(ws=need) == EWORKSPACE                                         # the two-launch form's workspace does not hold the slabs
    assert bwd1(hd=40) == EINVAL and bwd1(nseg=4, mlen=128) == EINVAL and bwd1(out=None) == EINVAL
    # the forward with the log-sum-exp given is the dropout pass only
    lse_fwd = lambda pd=0.25, seed=p(4), lse=p(3): lib.advmil_mha_fwd_lse(   # noqa: E731
        p(0), p(1), 512, 8, 48, 1, None, 512, pd, seed, 0, None, p(2), lse, None)
    assert lse_fwd(pd=0.0) == EINVAL and lse_fwd(seed=None) == EINVAL and lse_fwd(lse=None) == EINVAL


def test_optimizer_graph_and_evaluator_groups(L):
    lib = L.lib()
    p = lambda k: ctypes.c_void_p(A16 + (k << 20))   # noqa: E731
    assert lib.advmil_adam_step(p(0), p(1), p(2), p(3), p(4), -5, 1e-3, 0.9, 0.999, 1e-8, 1.0, 0.0, p(5), None, None, 1, None, 0, None) == EINVAL
    assert lib.advmil_genconv_fwd(p(0), p(1), p(2), p(3), 1e-7, 100, 0, p(4), p(5), p(6), None) == EINVAL
    assert lib.advmil_cindex_counts(p(0), p(1), p(2), -1, 1e-8, p(3), None) == EINVAL
    assert lib.advmil_gan_d_loss(p(0), 4, None, None, 0, 9, 0.25, 0.0, p(1), p(2), None, None) == EINVAL                 # unknown loss kind
    assert lib.advmil_skinny_linear_fwd(p(0), p(1), None, 4, 64, 128, 0, p(2), None) == EINVAL                           # neither width is 1
    assert lib.advmil_stamp_clock(None, None) == EINVAL
    assert lib.advmil_seg_scale_rows(p(0), p(1), None, 64, 126, p(2), None) == EINVAL                                    # D not a multiple of 4
    assert lib.advmil_genconv_bwd(p(0), p(1), p(2), p(3), p(4), p(5), p(6), 1e-7, 100, 128, p(7), p(8), p(9), 0, None) == EINVAL      # workspace too small
    assert lib.advmil_genconv_fwd(p(0), p(1), p(2), p(3), 1e-7, 100, 128, p(4), p(5), None, None) == EINVAL                  # lse without agg
    assert lib.advmil_ln_relu_mean16_bwd(p(0), p(1), p(2), p(3), p(4), p(5), 64, 128, None, p(6), p(7), 0, None, None, None, 1, p(8), 1 << 20, None) == EINVAL   # neither dy nor its planes
    assert lib.advmil_ln_relu_mean16_bwd(p(0), p(1), p(2), p(3), p(4), p(5), 64, 128, None, p(6), p(7), 0, None, p(9), None, 1, p(8), 1 << 20, None) == EINVAL   # one plane only
