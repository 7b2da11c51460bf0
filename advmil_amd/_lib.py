"""ctypes binding of libadvmil_hip.so (C ABI declared in include/advmil_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or a
tensor is not on a HIP device, the ops raise. Build with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C advmil_amd/csrc`.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

# ADVMIL_HIP_LIB: load another build of the same library (kernel A/B probes under tools/probe); there is still no fallback
LIB_PATH = os.environ.get("ADVMIL_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libadvmil_hip.so")


class HipLibraryMissing(RuntimeError):
    pass


class AdvmilHipError(RuntimeError):
    pass


class Epilogue(ctypes.Structure):
    """advmil_epilogue_t"""
    _fields_ = [("bias", c_void_p), ("act0", c_int), ("act1", c_int), ("act_split", c_int), ("drop_p", c_float),
                ("seed", c_void_p), ("stream_id", c_uint64), ("rowv", c_void_p), ("colv", c_void_p), ("rowseg", c_void_p),
                ("maskref", c_void_p), ("ldmask", c_int), ("mask_scale", c_float), ("accumulate", c_int),
                ("alpha", c_float), ("a_hi", c_void_p), ("a_lo", c_void_p), ("b_hi", c_void_p), ("b_lo", c_void_p),
                ("c_hi", c_void_p), ("c_lo", c_void_p), ("gate_wc", c_void_p), ("gate_out", c_void_p), ("gate_np", c_int),
                ("rng_row", c_void_p), ("c2", c_void_p), ("ldc2", c_int64), ("n_split", c_int64), ("bias2", c_void_p), ("colsum", c_void_p), ("maskbits", c_void_p), ("ldbits", c_int64),
                ("gate_bits_a", c_void_p), ("gate_bits_b", c_void_p), ("ldgbits", c_int64), ("c_rows_pair32", c_int)]


class DenseLayer(ctypes.Structure):
    """advmil_dense_layer_t"""
    _fields_ = [("W", c_void_p), ("bias", c_void_p), ("dW", c_void_p), ("dbias", c_void_p), ("y", c_void_p), ("K", c_int), ("N", c_int),
                ("act", c_int), ("drop_p", c_float), ("stream_id", c_uint64)]


TAIL_MAXL = 3


class DTail(ctypes.Structure):
    """advmil_dtail_t"""
    _fields_ = [("B", c_int), ("nx", c_int), ("ny", c_int), ("prj_src", c_int), ("xin", c_void_p), ("tin", c_void_p),
                ("x", DenseLayer * TAIL_MAXL), ("y", DenseLayer * TAIL_MAXL), ("u", c_void_p), ("w_prj", c_void_p), ("b_prj", c_void_p),
                ("dw_prj", c_void_p), ("db_prj", c_void_p), ("seed", c_void_p), ("rng_row", c_void_p), ("out", c_void_p),
                ("dout", c_void_p), ("dxin", c_void_p), ("dtin", c_void_p), ("du", c_void_p)]


class GemmTnCall(ctypes.Structure):
    """advmil_gemm_tn_call_t"""
    _fields_ = [("M", c_int64), ("N", c_int64), ("K", c_int64), ("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64),
                ("C", c_void_p), ("ldc", c_int64), ("accumulate", ctypes.c_int32)]


class GHead(ctypes.Structure):
    """advmil_ghead_t"""
    _fields_ = [("B", ctypes.c_int32), ("d0", ctypes.c_int32), ("d1", ctypes.c_int32), ("d2", ctypes.c_int32), ("noise_mode", ctypes.c_int32),
                ("out_act", ctypes.c_int32), ("x", c_void_p), ("ldx", c_int64), ("Wr", c_void_p), ("br", c_void_p), ("W0", c_void_p),
                ("b0", c_void_p), ("W1", c_void_p), ("b1", c_void_p), ("p1", c_float), ("p2", c_float), ("seed", c_void_p),
                ("sid1", c_uint64), ("sid2", c_uint64), ("sid_noise", c_uint64), ("rng_row", c_void_p), ("noise", c_void_p),
                ("hs", c_void_p), ("h2", c_void_p), ("pred", c_void_p), ("dpred", c_void_p), ("dx", c_void_p), ("lddx", c_int64),
                ("dWr", c_void_p), ("dbr", c_void_p), ("dW0", c_void_p), ("db0", c_void_p), ("dW1", c_void_p), ("db1", c_void_p),
                ("ws", c_void_p), ("ws_bytes", c_size_t)]


# name -> (restype, argtypes); must list every symbol include/advmil_hip.h declares
SIGNATURES = {
    "advmil_version": (c_int, []),
    "advmil_gemm_f32_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "advmil_gemm_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                c_int64, ctypes.POINTER(Epilogue), c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_split_planes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "advmil_stage_bag": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_gate_interleave": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "advmil_gate_partial_sum": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "advmil_gemm_f32_gate_blocks": (c_int, [c_int, c_int64]),
    "advmil_gemm_f32_colsum_rows": (c_int64, [c_int, c_int64, c_int64]),
    "advmil_merge_partials": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "advmil_set_gemm_mode": (c_int, [c_int]),
    "advmil_get_gemm_mode": (c_int, []),
    "advmil_gemm_f32_plan": (c_int, [c_int64, c_int64, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "advmil_gemm_f32_plan_layout": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "advmil_gemm_f32_plan_planes": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, ctypes.POINTER(c_int)]),
    "advmil_gemm_f32_plan_tn_planes": (c_int, [c_int64, c_int64, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "advmil_gemm_f32_tiled": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                      c_int64, ctypes.POINTER(Epilogue), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_mha_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_float, c_void_p, c_uint64,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_mha_fwd_lse": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_float, c_void_p, c_uint64,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_mha_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "advmil_mha_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                               c_float, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_mha_bwd1_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int64]),
    "advmil_mha_bwd1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                                c_float, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_add_dropout_ln_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_float, c_void_p,
                                          c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_add_dropout_ln_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_add_dropout_ln_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p,
                                          c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                          c_void_p]),
    "advmil_gate_score_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_uint64, c_uint64, c_int64, c_int64,
                                      c_void_p, c_void_p, c_void_p]),
    "advmil_softmax_pool_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "advmil_softmax_pool_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_softmax_pool_mean_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "advmil_softmax_pool_mean_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_softmax_pool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p,
                                        c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_softmax_pool_fwd_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p,
                                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_softmax_pool_bwd_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p,
                                               c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_dropout_planes": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "advmil_gate_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_gate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_uint64, c_uint64, c_int64, c_int64,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                c_void_p]),
    "advmil_colsum_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_act_dropout_bwd": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_uint64, c_int64, c_int64, c_void_p,
                                       c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_colsum": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_ln_relu_mean16_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "advmil_ln_relu_mean16_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_ln_relu_mean16_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_ln_relu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_ln_relu_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_ln_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_genconv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "advmil_genconv_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "advmil_genconv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64,
                                   c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                 c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "advmil_adam_blocks": (c_int, [c_int64]),
    "advmil_step_seed_tick": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "advmil_abs_sum": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "advmil_abs_sum_workspace_bytes": (c_size_t, [c_int64]),
    "advmil_uniform_fill": (c_int, [c_void_p, c_int64, c_void_p, c_uint64, c_void_p, c_int64, c_void_p]),
    "advmil_dropout_apply": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_uint64, c_void_p, c_int64, c_void_p]),
    "advmil_seed_advance": (c_int, [c_void_p, c_uint64, c_void_p]),
    "advmil_seg_scale_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "advmil_stamp_clock": (c_int, [c_void_p, c_void_p]),
    "advmil_clock_rate_khz": (c_int64, []),
    "advmil_small_linear_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_uint64,
                                        c_void_p, c_void_p, c_void_p]),
    "advmil_small_linear_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "advmil_small_linear_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                        c_uint64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "advmil_skinny_linear_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "advmil_skinny_linear_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p]),
    "advmil_prj_head_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "advmil_prj_head_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p]),
    "advmil_gan_d_loss": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_gan_g_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_float, c_float,
                                  c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_dx_chain_fwd": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_dx_chain_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "advmil_dx_chain_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "advmil_dx_chain_bwd": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_float, c_float, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "advmil_dtail_fwd": (c_int, [ctypes.POINTER(DTail), c_void_p]),
    "advmil_dtail_bwd": (c_int, [ctypes.POINTER(DTail), c_void_p]),
    "advmil_gemm_tn_group_workspace_bytes": (c_size_t, [ctypes.POINTER(GemmTnCall), c_int]),
    "advmil_gemm_tn_group": (c_int, [ctypes.POINTER(GemmTnCall), c_int, c_void_p, c_size_t, c_void_p]),
    "advmil_ghead_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "advmil_ghead_fwd": (c_int, [ctypes.POINTER(GHead), c_void_p]),
    "advmil_ghead_bwd": (c_int, [ctypes.POINTER(GHead), c_void_p]),
    "advmil_defer_sums": (c_int, [c_void_p, c_int]),
    "advmil_flush_sums": (c_int, [c_void_p]),
    "advmil_pending_sums": (c_int, [c_void_p]),
    "advmil_cindex_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
}

_lib = None
DEFAULT_GEMM_MODE = "bf16x3"
_GEMM_MODES = {"exact": 0, "f32": 0, "bf16x3": 1, "split": 1}


def lib():
    """Load (once) and return the shared library; raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: the AdvMIL HIP kernels are not built (run __graft_entry__.build()). "
                "There is no CPU fallback in the product path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        # Arithmetic of the contraction engine: "bf16x3" (the product default -- the mode bench.py's `value` is quoted in; the whole
        # golden suite passes in it at the same 2e-5 as exact) | "exact" (fp32 MFMA, ~0.46 x the rate). cfg['gemm_mode'] overrides it per handler.
        mode = os.environ.get("ADVMIL_GEMM_MODE") or DEFAULT_GEMM_MODE
        if mode not in _GEMM_MODES:
            raise ValueError(f"ADVMIL_GEMM_MODE={mode!r}: expected one of {sorted(_GEMM_MODES)}")
        handle.advmil_set_gemm_mode(_GEMM_MODES[mode])
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        kind = {-1: "invalid argument (shape/alignment/null)", -2: "workspace too small"}.get(code, f"hipError_t {code}")
        raise AdvmilHipError(f"{what} failed: {kind}")
