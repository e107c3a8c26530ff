"""ctypes binding of libmirror_hip.so (C ABI declared in include/mirror_hip.h).

The product path has no fallback: if the shared library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRROR_HIP_LIB: another build of the same ABI (A/B timing of kernel variants inside one process launch)
LIB_PATH = os.environ.get("MIRROR_HIP_LIB") or os.path.join(_HERE, "lib", "libmirror_hip.so")

MH_F32, MH_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class GemmEpi(C.Structure):
    """mh_gemm_epi of include/mirror_hip.h (field order = the header's)."""
    _fields_ = [
        ("kind", C.c_int32), ("resid", C.c_void_p),
        ("p", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64), ("dev_base", C.c_void_p),
        ("mask", C.c_void_p), ("token", C.c_void_p), ("pos", C.c_void_p),
        ("rows_per_batch", C.c_int32), ("first", C.c_int32),
        ("tgt", C.c_void_p), ("tgt_bs", C.c_int64), ("sq", C.c_void_p),
    ]


class SkinnyWgradItem(C.Structure):
    """mh_skinny_wgrad_item of include/mirror_hip.h."""
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("dw", C.c_void_p), ("lddw", C.c_int64),
                ("db", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("dt_dy", C.c_int32), ("dt_x", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("a_kc", C.c_int32), ("b_kc", C.c_int32),
        ("dtA", C.c_int32), ("dtB", C.c_int32), ("dtC", C.c_int32), ("mma", C.c_int32),
        ("batch1", C.c_int32), ("batch2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64),
        ("alpha", C.c_float), ("diag", C.c_float),
        ("act", C.c_int32), ("accumulate", C.c_int32), ("split_k", C.c_int32),
        ("R", C.c_void_p), ("rcoef", C.c_float),
        ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
        ("C2", C.c_void_p), ("r_bf16", C.c_int32),
        ("k_segments", C.c_int32), ("sA_seg", C.c_int64), ("sB_seg", C.c_int64),
        ("row_softmax", C.c_int32),
        ("epi", C.POINTER(GemmEpi)), ("a_rows_per_batch", C.c_int32), ("a_row_skip", C.c_int32),
        ("shared_chip", C.c_int32),
        ("c_rows_per_batch", C.c_int32), ("c_row_skip", C.c_int32),
        ("window_batches", C.c_int32),
    ]


P, I, L, F, U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64


class LossTermsDesc(C.Structure):
    """mh_loss_terms of include/mirror_hip.h (field order = the header's)."""
    _fields_ = [
        ("wsi_emb", C.c_void_p), ("rna_emb", C.c_void_p), ("logit_scale", C.c_void_p), ("align_ext", C.c_void_p),
        ("B", C.c_int32), ("D", C.c_int32), ("has_align", C.c_int32),
        ("rna_pred", C.c_void_p), ("rna_tgt", C.c_void_p), ("rna_mask", C.c_void_p), ("n_rna", C.c_int64),
        ("w_mu", C.c_void_p), ("w_logstd", C.c_void_p), ("r_mu", C.c_void_p), ("r_logstd", C.c_void_p),
        ("n_wstyle", C.c_int64), ("n_rstyle", C.c_int64), ("rows_wstyle", C.c_int32), ("rows_rstyle", C.c_int32),
        ("w_score", C.c_void_p), ("r_score", C.c_void_p), ("Bc", C.c_int32), ("P", C.c_int32),
        ("wsi_acc", C.c_void_p), ("weight", C.c_float * 6),
        ("scratch", C.c_void_p), ("save", C.c_void_p), ("out", C.c_void_p),
        ("g_total", C.c_void_p), ("g_terms", C.c_void_p),
        ("d_wsi_emb", C.c_void_p), ("d_rna_emb", C.c_void_p), ("d_logit_scale", C.c_void_p), ("d_align_ext", C.c_void_p),
        ("d_rna_pred", C.c_void_p), ("d_rna_tgt", C.c_void_p), ("d_w_mu", C.c_void_p), ("d_w_logstd", C.c_void_p), ("d_r_mu", C.c_void_p),
        ("d_r_logstd", C.c_void_p), ("d_w_score", C.c_void_p), ("d_r_score", C.c_void_p),
    ]


class RnaBlockDesc(C.Structure):
    """mh_rna_block of include/mirror_hip.h (field order = the header's)."""
    _fields_ = (
        [("B", C.c_int32), ("D", C.c_int32), ("Hh", C.c_int32), ("H", C.c_int32), ("eps", C.c_float), ("p_drop", C.c_float),
         ("seed", C.c_uint64), ("offset", C.c_uint64), ("dev_base", C.c_void_p)]
        + [(n, C.c_void_p) for n in (
            "w_qkv", "w_proj", "w_fc1", "w_fc2", "wt_qkv", "wt_proj", "wt_fc1", "wt_fc2",
            "b_qkv", "b_proj", "b_fc1", "b_fc2", "g1", "be1", "g2", "be2",
            "x", "y", "stats", "qkv", "attn", "o", "x1", "u", "f", "dy", "dx",
            "dw_qkv", "dw_proj", "dw_fc1", "dw_fc2", "db_qkv", "db_proj", "db_fc1", "db_fc2", "dg1", "dbe1", "dg2", "dbe2",
            "scratch")])


# name -> argtypes (the trailing stream argument is appended automatically)
_SIGS = {
    "mh_gemm": [C.POINTER(GemmDesc)],
    "mh_skinny_fwd": [P, L, P, L, P, P, L, P, L, I, I, I, I, I, I],
    "mh_skinny_wgrad": [P, L, P, L, P, L, P, I, I, I, I, I, I],
    "mh_skinny_wgrad_many": [C.POINTER(SkinnyWgradItem), I],
    "mh_transpose_bf16": [P, P, I, I],
    "mh_transpose_bf16_many": [P, P, P, I, I, I, I],
    "mh_layernorm_fwd": [P, P, P, P, P, P, I, I, I, L, L, F, I, I],
    "mh_layernorm_fwd_dual": [P, P, P, P, P, P, P, I, I, I, L, L, F],
    "mh_layernorm_fwd_q8": [P, P, P, P, P, P, I, I, I, L, L, F, P, P, P, F, P],
    "mh_layernorm_bwd": [P, P, P, P, P, P, P, P, I, I, I, L, L, I, I, I, I, P, L],
    "mh_layernorm_bwd_fan": [P, P, P, P, P, P, P, P, I, I, I, L, L, I, I, P, L, P, F, P],
    "mh_layernorm_bwd_drop": [P, P, P, P, P, P, P, P, I, I, I, L, L, I, I, P, L, P, F, P, P, F, U64, U64, P, P, I],
    "mh_layernorm_bwd_lm": [P, P, P, P, P, P, P, P, I, I, I, L, L, I, I, I, I, P, L, P, I, I, P, I, I, P, P, P],
    "mh_layernorm_fwd_lm": [P, P, P, P, P, P, P, P, I, I, I, L, I, I, F, P, P],
    "mh_softmax_fwd": [P, P, L, I, L, L, I, I],
    "mh_softmax_bwd": [P, P, P, L, I, L, L, L, I, I, I],
    "mh_landmark_fwd": [P, P, I, I, I, I, I],
    "mh_landmark_bwd": [P, P, I, I, I, I, I],
    "mh_resconv_fwd": [P, L, L, P, P, L, L, I, I, I, I, I, I, I, I, I],
    "mh_resconv_wgrad": [P, L, L, P, L, L, P, I, I, I, I, I, I, I],
    "mh_resconv_bwd": [P, L, L, P, L, L, P, P, L, L, P, P, L, I, I, I, I, I, I, I],
    "mh_pinv_absmax": [P, P, I, I],
    "mh_pinv_z0": [P, P, P, I, I],
    "mh_pinv_z0_bwd": [P, P, P, P, P, P, I, I, I],
    "mh_pinv_s2_bwd": [P, P, P, P, P, I, I, I, P, I],
    "mh_eye_minus": [P, P, F, I, I],
    "mh_pinv_chain_prep": [P, P, P, P, P, I, I],
    "mh_pinv_chain_pack": [P, P, I, I],
    "mh_pinv_chain_fwd": [P, P, P, I, I, I, P, P, I],
    "mh_nys_sim2": [P, P, P, P, P, I, I, I, I, F, L, P],
    "mh_nys_dz_dav": [P, P, P, P, P, P, I, I, I],
    "mh_pinv_chain_bwd": [P, P, P, P, P, P, I, I, I],
    "mh_nys_attn1_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, F, I, L, P],
    "mh_nys_attn1_fwd_q8": [P, P, P, P, P, I, I, I, I, I, F, I, P, P, P, F, P, P],
    "mh_nys_attn3_fwd": [P, P, P, P, P, L, P, P, I, I, I, I, I, F, L, P, P],
    "mh_nys_attn1_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, L, I],
    "mh_nys_attn3_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, L, I],
    "mh_seq_finish": [P, P, I, I, I, I, I],
    "mh_seq_finish_bwd": [P, P, I, I, I, I, I],
    "mh_ppeg_merge": [P, P, P, P, P, P, P, P, I],
    "mh_ppeg_fwd": [P, P, P, P, I, I, I, I, I, I],
    "mh_ppeg_wgrad": [P, P, P, P, I, I, I, I, I],
    "mh_ppeg_grad_scatter": [P, P, P, P, P, P, P, P, I],
    "mh_rank_mask": [P, P, I, I, I],
    "mh_mask_apply_fwd": [P, P, P, P, P, I, I, I, I, I, I, I],
    "mh_mask_apply_bwd": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "mh_headattn_fwd": [P, P, P, I, I, I, I],
    "mh_headattn_bwd": [P, P, P, P, I, I, I, I],
    "mh_add": [P, P, P, L, I, I, I],
    "mh_lm_merge": [P, P, P, L, I, L, I],
    "mh_cast": [P, P, L, I, I],
    "mh_gelu_fwd": [P, P, L, I, I],
    "mh_gelu_bwd": [P, P, P, L, I, I, I],
    "mh_relu_bwd": [P, P, P, L, I, L, L, L, I, I, I],
    "mh_dropout": [P, P, L, F, U64, U64, P, I, I],
    "mh_noise_draws": [P, L, L, U64, U64, P],
    "mh_gemm_w4": [P, P, P, P, I, I, I, L, L, L],
    "mh_dropout_add": [P, P, P, L, F, U64, U64, P, I],
    "mh_dropout_lite": [P, P, P, L, F, U64, U64, P, I, I],
    "mh_timestamp": [P],
    "mh_dropout_lite_colsum": [P, P, L, I, F, U64, U64, P, P],
    "mh_colsum": [P, P, L, I, L, I],
    "mh_l2norm_fwd": [P, P, P, I, I, L, F, I, I],
    "mh_l2norm_bwd": [P, P, P, P, I, I, L, F, I, I, I, I],
    "mh_exp_fwd": [P, P, L],
    "mh_exp_bwd": [P, P, P, L, I],
    "mh_reparam_fwd": [P, P, P, P, L],
    "mh_reparam_bwd": [P, P, P, P, P, P, P, L],
    "mh_ce_rows_fwd": [P, L, P, F, I, I, I, F, P, P, P],
    "mh_ce_rows_bwd": [P, L, P, F, P, P, I, F, P, P, I, I, I],
    "mh_mse_masked_fwd": [P, P, P, P, L, I, L, L, I, I],
    "mh_mse_masked_bwd": [P, P, P, P, P, F, P, P, L, I, L, L, I, I, I, P, I],
    "mh_fanout_bwd": [P, P, F, P, P, I, I, I, I],
    "mh_gather_rows": [P, P, P, L, L, L, I],
    "mh_quant_fp8": [P, L, P, P, P, I],
    "mh_quant_fp8_delayed": [P, L, P, P, P, P, F, I],
    "mh_gemm_fp8": [P, L, L, P, L, P, L, L, I, P, P, P, I, I, I, I, I],
    "mh_weighted_sum": [P, P, P, P, P, P, F, F, F, F, F, F, I, P],
    "mh_weighted_sum_bwd": [P, F, F, F, F, F, F, I, P],
    "mh_softmax_masked_fwd": [P, P, P, P, L, I, I, I, I, I],
    "mh_softmax_masked_bwd": [P, P, P, P, P, L, I, I, I, I, I],
    "mh_row_scale": [P, P, P, L, I, I],
    "mh_keymask_plan": [P, P, P, P, L, L, I, I, I, I],
    "mh_kl_fwd": [P, P, P, L, F],
    "mh_kl_bwd": [P, P, P, P, P, L, F],
    "mh_symkl_fwd": [P, P, P, I, I, F],
    "mh_symkl_bwd": [P, P, P, P, P, I, I, F],
    "mh_rownorm_": [P, P, I, I, F],
    "mh_clamp_": [P, L, F, F],
    "mh_adam": [P, P, P, P, P, L, F, F, F, F, F, F, F, P, L, F, F, P, L, I, L, L],
    "mh_grad_clip": [P, L, F, F, P, P],
    "mh_rna_block_fwd": [C.POINTER(RnaBlockDesc)],
    "mh_rna_block_bwd": [C.POINTER(RnaBlockDesc)],
    "mh_loss_terms_fwd": [C.POINTER(LossTermsDesc)],
    "mh_loss_terms_bwd": [C.POINTER(LossTermsDesc)],
}
EXPORTS = sorted(list(_SIGS) + ["mh_last_error", "mh_version", "mh_exp_build", "mh_gemm_select_pp", "mh_gemm_variant_name", "mh_device_ok", "mh_nys_attn3_ws_floats", "mh_rna_block_workspace_bytes",
                                 "mh_gemm_workspace_bytes", "mh_layernorm_bwd_workspace_bytes", "mh_nys_attn3_workspace_bytes",
                                 "mh_pinv_chain_workspace_bytes", "mh_resconv_bwd_workspace_bytes", "mh_mask_apply_bwd_dbias_ok"])

_lib = None

# The ABI generation this binding was written against (mh_version() of csrc/errors.cpp).  _SIGS above restates the argument lists of
# include/mirror_hip.h by hand: a library built from another generation would be called with shifted arguments (a stream where a
# counter belongs) and corrupt device memory silently, so load() refuses anything but this exact number.
ABI_VERSION = 118


class MirrorHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (once). Raises MirrorHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirrorHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mirror_amd/csrc` — mirror_amd has no CPU/PyTorch fallback path")
    lib = C.CDLL(LIB_PATH)
    lib.mh_last_error.restype = C.c_char_p
    lib.mh_last_error.argtypes = []
    lib.mh_version.restype = C.c_int
    lib.mh_version.argtypes = []
    got = int(lib.mh_version())
    if got != ABI_VERSION:
        raise MirrorHipError(
            f"{LIB_PATH} is ABI v{got}, this binding (mirror_amd/_lib.py) is v{ABI_VERSION}: the argument lists differ — rebuild the "
            "library (`make -C mirror_amd/csrc`, or __graft_entry__.build()); the same holds for a library named by MIRROR_HIP_LIB")
    lib.mh_exp_build.restype = C.c_int
    lib.mh_gemm_variant_name.restype = C.c_char_p
    lib.mh_gemm_variant_name.argtypes = []
    lib.mh_gemm_select_pp.restype = C.c_int
    lib.mh_gemm_select_pp.argtypes = [C.c_int]
    lib.mh_device_ok.restype = C.c_int
    lib.mh_nys_attn3_ws_floats.restype = C.c_int64
    lib.mh_nys_attn3_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.mh_gemm_workspace_bytes.restype = C.c_int64
    lib.mh_gemm_workspace_bytes.argtypes = [C.POINTER(GemmDesc)]
    lib.mh_layernorm_bwd_workspace_bytes.restype = C.c_int64
    lib.mh_layernorm_bwd_workspace_bytes.argtypes = [C.c_int64, C.c_int]
    lib.mh_nys_attn3_workspace_bytes.restype = C.c_int64
    lib.mh_nys_attn3_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.mh_pinv_chain_workspace_bytes.restype = C.c_int64
    lib.mh_pinv_chain_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mh_resconv_bwd_workspace_bytes.restype = C.c_int64
    lib.mh_resconv_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mh_mask_apply_bwd_dbias_ok.restype = C.c_int
    lib.mh_mask_apply_bwd_dbias_ok.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.mh_rna_block_workspace_bytes.restype = C.c_int64
    lib.mh_rna_block_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = list(sig) + [C.c_void_p]
        fn.restype = C.c_int
    _lib = lib
    return lib


_fns: dict = {}


# timing experiment (tools/exp/skip_cost.sh): entry points named in MH_EXP_SKIP return at once — the results are garbage, the
# step time shows what each of them costs INSIDE the step (on its stream, beside whatever it overlaps) rather than alone
# Both this switch and MH_EXP_CHAIN_SKIP only exist in a library built with `make EXP=1` (mh_exp_build() == 1); with the
# shipped build a set MH_EXP_* variable is an error, not a silent fake number.
_SKIP = None


def _exp_skip() -> frozenset:
    global _SKIP
    if _SKIP is None:
        names = [k for k in os.environ if k.startswith("MH_EXP_")]
        lib = load()
        exp_build = bool(lib.mh_exp_build())
        if names and not exp_build:
            raise MirrorHipError(f"{', '.join(sorted(names))} set, but libmirror_hip.so is not a timing-experiment build "
                                 "(make -C mirror_amd/csrc EXP=1): refusing to run with result-corrupting switches")
        if names:
            import warnings
            warnings.warn(f"mirror_amd: timing-experiment switches active ({', '.join(sorted(names))}): RESULTS ARE GARBAGE")
        _SKIP = frozenset(x for x in os.environ.get("MH_EXP_SKIP", "").split(",") if x)
    return _SKIP


def call(name: str, *args, stream: int = 0) -> None:
    skip = _SKIP if _SKIP is not None else _exp_skip()
    if skip and name in skip:
        return
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args, stream)
    if rc != 0:
        raise MirrorHipError(f"{name} failed ({rc}): {load().mh_last_error().decode()}")
