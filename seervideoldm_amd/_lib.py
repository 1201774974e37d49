"""ctypes binding of libseer_hip.so (the C ABI declared in include/seer_hip.h).

The library is the product: there is NO fallback.  If it is missing, `load()` raises with the build command;
if a call returns a negative code, `check()` raises `SeerHipError` with the entry point's name.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_LIB_PATH = _PKG / "lib" / "libseer_hip.so"
_lib = None

ABI_VERSION = 24

SEER_GEMM_PLAIN = 0
SEER_GEMM_CONV3X3 = 1
SEER_EPI_GEGLU = 1
SEER_EPI_OUT_F32 = 2
SEER_EPI_SILU = 4
SEER_EPI_TRANS_OUT = 8
SEER_EPI_ROTARY = 16
SEER_EPI_COLSCALE = 32
SEER_EPI_F16 = 64
SEER_ENOSYS = -38
SEER_DT_BF16, SEER_DT_F16 = 0, 1
SEER_ATTN_Q_PRESCALED = 1
SEER_ATTN_F16 = 2
SEER_TILE_AUTO, SEER_TILE_128x128, SEER_TILE_64x64, SEER_TILE_128x64 = 0, 1, 2, 3


class SeerHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("rowvec", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K1", C.c_int32),
        ("lda", C.c_int32), ("lda2", C.c_int32), ("ldr", C.c_int32), ("ldc", C.c_int32),
        ("rows_per_batch", C.c_int32), ("rowvec_ld", C.c_int32),
        ("mode", C.c_int32), ("epilogue", C.c_uint32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32),
        ("batch", C.c_int32),
        ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64),
        ("tile", C.c_int32), ("splits", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("rot_table", C.c_void_p), ("rot_tokens_per_batch", C.c_int32), ("rot_pos_offset", C.c_int32),
        ("rot_head_dim", C.c_int32), ("rot_dim", C.c_int32), ("rot_cols", C.c_int32),
        ("pad_after_only", C.c_int32),
        ("col_scale_cols", C.c_int32), ("col_scale", C.c_float),
        ("colsum", C.c_void_p), ("sync", C.c_void_p), ("sync_bytes", C.c_int64),
        ("colsum_fx", C.c_void_p), ("colsum_fx_rows", C.c_int32), ("colsum_fx_reps", C.c_int32),
        ("rowstat", C.c_void_p), ("ln_rowstat", C.c_void_p), ("ln_wsum", C.c_void_p), ("ln_eps", C.c_float),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64), ("o_bs", C.c_int64),
        ("q_ss", C.c_int32), ("k_ss", C.c_int32), ("v_ss", C.c_int32), ("o_ss", C.c_int32),
        ("batch", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("Sq", C.c_int32), ("Sk", C.c_int32), ("causal", C.c_int32), ("scale", C.c_float),
        ("window_ws", C.c_int32), ("F", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Fq", C.c_int32), ("causal_offset", C.c_int32),
        ("lse", C.c_void_p),
        ("flags", C.c_uint32), ("variant", C.c_int32),
        ("q_hs", C.c_int64), ("k_hs", C.c_int64), ("v_hs", C.c_int64),
    ]


class TnItem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("colsum", C.c_void_p),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("reserved", C.c_int32)]


class ColfinalItem(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p),
                ("nblocks", C.c_int32), ("NV", C.c_int32), ("C", C.c_int32), ("reserved", C.c_int32)]


class RowChainDesc(C.Structure):
    _fields_ = [
        ("inp", C.c_void_p), ("ld_in", C.c_int32),
        ("gn_stats", C.c_void_p), ("gn_fx", C.c_void_p), ("gn_fx_reps", C.c_int32), ("gn_count", C.c_double), ("gn_eps", C.c_float), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),
        ("rows_per_batch", C.c_int64), ("groups", C.c_int32),
        ("w1f", C.c_void_p), ("b1", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_int32), ("h", C.c_void_p), ("ldh", C.c_int32),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
        ("w2f", C.c_void_p), ("n2", C.c_int32), ("out", C.c_void_p), ("ldo", C.c_int32),
        ("col_scale", C.c_float), ("scale_thirds", C.c_int32),
        ("rot_table", C.c_void_p), ("rot_tokens_per_batch", C.c_int32), ("rot_pos_offset", C.c_int32), ("rot_head_dim", C.c_int32),
        ("rot_dim", C.c_int32), ("rot_thirds", C.c_int32),
        ("M", C.c_int64), ("dtype", C.c_int32),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("fwd", AttnDesc),
        ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("do_bs", C.c_int64), ("dq_bs", C.c_int64), ("dk_bs", C.c_int64), ("dv_bs", C.c_int64),
        ("do_ss", C.c_int32), ("dq_ss", C.c_int32), ("dk_ss", C.c_int32), ("dv_ss", C.c_int32),
        ("delta", C.c_void_p),
    ]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> argtypes; every symbol include/seer_hip.h declares (tests/test_abi.py checks the list against the header)
SIGNATURES = {
    "seer_abi_version": ([], C.c_int),
    "seer_strerror": ([C.c_int], C.c_char_p),
    "seer_build_arch": ([], C.c_char_p),
    "seer_gemm_bf16": ([C.POINTER(GemmDesc), _vp], C.c_int),
    "seer_gemm_workspace_bytes": ([C.POINTER(GemmDesc)], C.c_int64),
    "seer_gemm_colsum_rows": ([C.POINTER(GemmDesc)], C.c_int32),
    "seer_gemm_sync_bytes": ([C.POINTER(GemmDesc)], C.c_int64),
    "seer_gemm_rowstat_ok": ([C.POINTER(GemmDesc)], C.c_int32),
    "seer_gemm_lnfold_ok": ([C.POINTER(GemmDesc)], C.c_int32),
    "seer_attn_fwd": ([C.POINTER(AttnDesc), _vp], C.c_int),
    "seer_rotary_table": ([_vp, _i32, _i32, _vp, _vp], C.c_int),
    "seer_rotary_inplace": ([_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_groupnorm_workspace_floats": ([_i32, _i32, _i64, _i32], C.c_int64),
    "seer_groupnorm_stats": ([_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp], C.c_int),
    "seer_groupnorm_stats_from_colsums": ([_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_groupnorm_apply": ([_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _f64, _f32, _vp, _vp, _i32, _vp, _vp], C.c_int),
    "seer_groupnorm_apply_from_colsums": ([_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _f64, _f32, _vp, _vp,
                                           _i32, _vp, _vp], C.c_int),
    "seer_groupnorm_apply_fx": ([_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _f64, _f32, _vp, _vp, _i32, _vp, _vp, _vp], C.c_int),
    "seer_groupnorm_stats_fx": ([_vp, _i32, _i32, _i64, _vp, _i32, _vp], C.c_int),
    "seer_groupnorm_apply_from_colsums_dt": ([_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _f64, _f32, _vp, _vp,
                                              _i32, _vp, _i32, _vp], C.c_int),
    "seer_groupnorm_apply_fx_dt": ([_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _f64, _f32, _vp, _vp, _i32, _vp, _vp, _i32,
                                    _vp], C.c_int),
    "seer_ff_fused_c320": ([_vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp], C.c_int),
    "seer_ff_fused_c320_dt": ([_vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp],
                              C.c_int),
    "seer_rowchain_c320": ([C.POINTER(RowChainDesc), _vp], C.c_int),
    "seer_rowchain_pack": ([_vp, _i32, _i32, _vp, _vp], C.c_int),
    "seer_ff_fused_c320_pre": ([_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                _vp, _i32, _vp], C.c_int),
    "seer_ff_fused_pack_w1": ([_vp, _vp, _vp], C.c_int),
    "seer_ff_fused_pack_wcat": ([_vp, _vp, _vp], C.c_int),
    "seer_gemm_colsum_fx_layout": ([C.POINTER(GemmDesc), _i32, C.POINTER(C.c_int32)], C.c_int32),
    "seer_groupnorm_stats_dt": ([_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp], C.c_int),
    "seer_groupnorm_apply_dt": ([_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _f64, _f32, _vp, _vp, _i32, _vp, _i32, _vp], C.c_int),
    "seer_softmax_rows_dt": ([_vp, _i32, _i64, _i32, _i32, _f32, _vp, _i32, _i32, _vp], C.c_int),
    "seer_conv_in_dt": ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp], C.c_int),
    "seer_conv_out_dt": ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp], C.c_int),
    "seer_layernorm": ([_vp, _i64, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _vp], C.c_int),
    "seer_layernorm_dt": ([_vp, _i64, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _vp], C.c_int),
    "seer_softmax_rows": ([_vp, _i32, _i64, _i32, _i32, _f32, _vp, _i32, _vp], C.c_int),
    "seer_conv1x1_nchw_f32": ([_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp], C.c_int),
    "seer_timestep_embedding": ([_vp, _i32, _i32, _i32, _f32, _vp, _vp], C.c_int),
    "seer_linear_smallm": ([_vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_linear_smallm_dt": ([_vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp], C.c_int),
    "seer_conv_in": ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp], C.c_int),
    "seer_conv_out": ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp], C.c_int),
    "seer_cast_f32_bf16": ([_vp, _i64, _vp, _vp], C.c_int),
    "seer_cast_f32_dt": ([_vp, _i64, _vp, _i32, _vp], C.c_int),
    "seer_nchw_f32_to_nhwc_bf16": ([_vp, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_nhwc_bf16_to_nchw_f32": ([_vp, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_cfg_ddim_step": ([_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "seer_ddim_step_begin": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "seer_cfg_ddim_step_dev": ([_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "seer_clamp01": ([_vp, _i64, _vp], C.c_int),
    "seer_gaussian_sample": ([_vp, _i32, _i32, _i32, _vp, _vp, _vp], C.c_int),
    # training step
    "seer_attn_bwd": ([C.POINTER(AttnBwdDesc), _vp], C.c_int),
    "seer_gemm_tn_workspace_bytes": ([_i32, _i32, _i32], C.c_int64),
    "seer_gemm_tn_f32": ([_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp], C.c_int),
    "seer_gemm_tn_grouped_workspace_bytes": ([C.POINTER(TnItem), _i32], C.c_int64),
    "seer_gemm_tn_grouped_f32": ([C.POINTER(TnItem), _i32, _vp, _i64, _vp], C.c_int),
    "seer_transpose_bf16": ([_vp, _i64, _i32, _i32, _vp, _i64, _vp], C.c_int),
    "seer_transpose_batched_bf16": ([_vp, _i32, _i64, _vp], C.c_int),
    "seer_colsum_workspace_floats": ([_i64, _i32], C.c_int64),
    "seer_colsum_bf16": ([_vp, _i64, _i32, _i32, _vp, _vp, _vp], C.c_int),
    "seer_layernorm_bwd": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp, _f32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp], C.c_int),
    "seer_layernorm_bwd_slabs": ([_i64], C.c_int64),
    "seer_layernorm_bwd_partials": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp, _f32, _vp, _i32, _vp, _i32, _vp, _vp], C.c_int),
    "seer_colfinal_grouped": ([C.POINTER(ColfinalItem), _i32, _vp], C.c_int),
    "seer_groupnorm_bwd_workspace_floats": ([_i32, _i32, _i64, _i32], C.c_int64),
    "seer_groupnorm_bwd": ([_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _f64, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp], C.c_int),
    "seer_geglu_fwd": ([_vp, _i64, _i32, _i32, _vp, _i32, _vp], C.c_int),
    "seer_geglu_bwd": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp], C.c_int),
    "seer_add_bf16": ([_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _vp], C.c_int),
    "seer_sumpool2x_bf16": ([_vp, _i32, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_zero_insert2x_bf16": ([_vp, _i32, _i32, _i32, _i32, _vp, _vp], C.c_int),
    "seer_mse_loss_grad": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp], C.c_int),
    "seer_text_loss_grad": ([_vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp], C.c_int),
    "seer_conv_out_bwd": ([_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp], C.c_int),
    "seer_axpby_f32": ([_vp, _vp, _f32, _f32, _i64, _vp], C.c_int),
    "seer_sumsq_f32": ([_vp, _i64, _vp, _vp, _vp], C.c_int),
    "seer_adamw_step": ([_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _f32, _vp, _vp], C.c_int),
}


def lib_path() -> Path:
    return Path(os.environ.get("SEER_HIP_LIB", _LIB_PATH))


def load():
    """dlopen libseer_hip.so and bind every entry point; raises if the library is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise SeerHipError(
            f"{path} not found: the HIP extension is the only compute path of seervideoldm_amd. "
            "Build it with `python -m seervideoldm_amd.build` (hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(str(path))
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    ver = lib.seer_abi_version()
    if ver != ABI_VERSION:
        raise SeerHipError(f"libseer_hip.so ABI {ver} != binding ABI {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().seer_strerror(code).decode()
        raise SeerHipError(f"{what} failed: {msg} ({code})")
