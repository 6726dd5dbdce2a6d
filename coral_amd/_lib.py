"""ctypes binding of libcoral_amd.so (the C ABI declared in include/coral_amd.h).

There is no CPU fallback: importing this module without the built library raises, and
every wrapper raises `CoralAmdError` when a kernel call reports a failure.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libcoral_amd.so"


class CoralAmdError(RuntimeError):
    """Raised when the HIP library is missing or a call into it fails."""


class CaGemmDesc(C.Structure):
    """Mirror of `CaGemmDesc` in include/coral_amd.h."""

    _fields_ = [
        ("A", C.c_void_p),
        ("B", C.c_void_p),
        ("C", C.c_void_p),
        ("C2", C.c_void_p),
        ("R", C.c_void_p),
        ("bias", C.c_void_p),
        ("M", C.c_int32),
        ("N", C.c_int32),
        ("K", C.c_int32),
        ("a_layout", C.c_int32),
        ("b_layout", C.c_int32),
        ("lda", C.c_int64),
        ("ldb", C.c_int64),
        ("ldc", C.c_int64),
        ("ldr", C.c_int64),
        ("a_kseg", C.c_int32),
        ("b_kseg", C.c_int32),
        ("a_kseg_stride", C.c_int64),
        ("b_kseg_stride", C.c_int64),
        ("batch1", C.c_int32),
        ("batch2", C.c_int32),
        ("sA1", C.c_int64),
        ("sA2", C.c_int64),
        ("sB1", C.c_int64),
        ("sB2", C.c_int64),
        ("sC1", C.c_int64),
        ("sC2", C.c_int64),
        ("sR1", C.c_int64),
        ("sR2", C.c_int64),
        ("sBias1", C.c_int64),
        ("sBias2", C.c_int64),
        ("epilogue", C.c_int32),
        ("out_f32", C.c_int32),
        ("accumulate", C.c_int32),
        ("alpha", C.c_float),
        ("dropout_p", C.c_float),
        ("dropout_seed", C.c_uint64),
        ("a_colsum", C.c_void_p),
        ("a_colsum_ld", C.c_int64),
        ("c_row_index", C.c_void_p),
        ("c_row_mul", C.c_int64),
        ("c_split_n", C.c_int32),
        ("C_hi", C.c_void_p),
        ("ldc_hi", C.c_int64),
        ("a_scale", C.c_void_p),
        ("b_scale", C.c_void_p),
        ("a_row_scale", C.c_void_p),
        ("c_sumsq", C.c_void_p),
        ("c_stream_out", C.c_int32),
        ("C8", C.c_void_p),
        ("c8_scale", C.c_void_p),
        ("c8_amax", C.c_void_p),
        ("a_ln_gamma", C.c_void_p),
        ("a_ln_beta", C.c_void_p),
        ("a_ln_eps", C.c_float),
        ("xcd_balanced", C.c_int32),
    ]


class CaReduceDesc(C.Structure):
    """include/coral_amd.h: CaReduceDesc (one reduction of ca_reduce_rows_multi)."""
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("stride", C.c_int64), ("nparts", C.c_int32),
                ("n", C.c_int32), ("accumulate", C.c_int32)]


class CaDecodeLayer(C.Structure):
    """Mirror of `CaDecodeLayer` in include/coral_amd.h (one record per decoder layer, in DEVICE memory)."""

    FIELDS = ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "wq2", "bq2", "wo2", "bo2", "ln3_g", "ln3_b",
              "w1", "b1", "w2", "b2", "self_kv", "cross_kv")
    _fields_ = [(n, C.c_void_p) for n in FIELDS]


class CaDecodeDesc(C.Structure):
    """Mirror of `CaDecodeDesc` in include/coral_amd.h."""

    _fields_ = ([("layers", C.c_void_p)]
                + [(n, C.c_int32) for n in ("n_layers", "B", "d", "f", "H", "Te", "max_len", "V")]
                + [(n, C.c_void_p) for n in ("embed", "embed_pos", "lnf_g", "lnf_b")]
                + [("eps", C.c_float), ("logits", C.c_void_p), ("ld_logits", C.c_int64), ("suppress", C.c_void_p),
                   ("out", C.c_void_p), ("done", C.c_void_p), ("ids", C.c_void_p), ("ld_ids", C.c_int64),
                   ("tok", C.c_void_p), ("pos", C.c_void_p), ("klen", C.c_void_p), ("pad_id", C.c_int32),
                   ("eos_id", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("status", C.c_void_p),
                   ("cross_head_major", C.c_int32)])


DECODE_MAX_B = 16


def decode_ws_bytes(B, d, f, H, n_layers):
    """CA_DECODE_WS_BYTES of include/coral_amd.h."""
    return (8192 + n_layers * 16 * H * 4 + 16 * (7 * d + f) * 2 + 16 * H * 4 * 16 * 66 * 4 + 256 * 16 * 8 + 4096)


class CaAttnDesc(C.Structure):
    """Mirror of `CaAttnDesc` in include/coral_amd.h."""

    _fields_ = ([(n, C.c_void_p) for n in ("Q", "K", "V", "O", "dO", "dQ", "dK", "dV", "lse", "Dq", "klen")]
                + [(n, C.c_int64) for n in ("ldq", "ldk", "ldv", "ldo", "lddo", "lddq", "lddk", "lddv",
                                             "sqb", "skb", "svb", "sob", "sdob", "sdqb", "sdkb", "sdvb")]
                + [(n, C.c_int32) for n in ("B", "H", "Tq", "Tk", "hd", "Tqp", "causal")]
                + [("scale", C.c_float), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64)]
                + [("O8", C.c_void_p), ("o8_scale", C.c_void_p), ("o8_amax", C.c_void_p),
                   ("split_ws", C.c_void_p), ("split_ws_bytes", C.c_int64)])


class CaFp8RefreshTask(C.Structure):
    """Mirror of `CaFp8RefreshTask` in include/coral_amd.h."""

    _fields_ = ([(n, C.c_void_p) for n in ("x_bf16", "q_fp8", "q_fp8_t", "scale", "amax_next")]
                + [("rows", C.c_int32), ("cols", C.c_int32)])


FP8_GROUP_MAX = 8  # CA_FP8_GROUP_MAX
KMAJOR, MNMAJOR = 0, 1
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_GELU_RESIDUAL = 0, 1, 2, 3, 4

_vp, _i32, _i64, _f32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/coral_amd.h declares.
SIGNATURES = {
    "ca_version": (C.c_int, []),
    "ca_last_error": (C.c_char_p, []),
    "ca_device_count": (C.c_int, []),
    "ca_gemm_bf16_group": (C.c_int, [_vp, _i32, _vp]),
    "ca_gemm_bf16": (C.c_int, [C.POINTER(CaGemmDesc), _vp]),
    "ca_gemm_fp8": (C.c_int, [C.POINTER(CaGemmDesc), _vp]),
    "ca_quantize_fp8": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ca_quantize_fp8_delayed": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ca_fp8_amax_rotate": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _vp]),
    "ca_dropout_rows_fp8": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _u64, _vp]),
    "ca_quantize_fp8_transposed": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "ca_fp8_refresh_group": (C.c_int, [_vp, _i32, _vp]),
    "ca_gemm_force_kernel": (C.c_int, [C.c_int]),
    "ca_gemm_debug_general_epilogue": (C.c_int, [C.c_int]),
    "ca_debug_cu_hog": (C.c_int, [_i32, _i32, _i32, C.c_double, _vp]),
    "ca_gemm_set_compute_cus": (C.c_int, [C.c_int]),
    "ca_background_update_fits": (C.c_int, [C.POINTER(C.c_int32)]),
    "ca_prof_begin": (C.c_int, []),
    "ca_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ca_attn_fwd": (C.c_int, [C.POINTER(CaAttnDesc), _vp]),
    "ca_attn_bwd": (C.c_int, [C.POINTER(CaAttnDesc), _vp]),
    "ca_decode_attn_qproj": (C.c_int, [C.POINTER(CaAttnDesc), _vp, _i64, _vp, _vp, _f32, _vp, _i64, _vp, _i32, _vp]),
    "ca_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "ca_layernorm_fwd_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _i32, _i32, _vp]),
    "ca_layernorm_bwd_ex": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    ),
    "ca_layernorm_fwd_fp8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "ca_layernorm_bwd_partial_floats": (_i64, [_i64, _i32]),
    "ca_layernorm_bwd": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    ),
    "ca_colsum_partial_floats": (_i64, [_i64, _i32]),
    "ca_colsum_bf16": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ca_dgelu_mul": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "ca_dropout_bf16": (C.c_int, [_vp, _vp, _i64, _f32, _u64, _vp]),
    "ca_reduce_rows_f32": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i32, _vp]),
    "ca_reduce_rows_multi": (C.c_int, [C.POINTER(CaReduceDesc), _i32, _vp]),
    "ca_wave_normalize": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _f32, _vp]),
    "ca_frame_lengths": (C.c_int, [_vp, _i32, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i32, _vp, _vp]),
    "ca_wave_scale": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp]),
    "ca_fir_filter": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i64, _vp]),
    "ca_mix_noise": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _vp]),
    "ca_white_noise": (C.c_int, [_vp, _i64, C.c_uint64, _vp]),
    "ca_pcm_prepare": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp]),
    "ca_conv0_ln_gelu_fwd": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _f32, _vp],
    ),
    "ca_conv0_bwd_partial_floats": (_i64, [_i32, _i64, _i32, _i32, _i32]),
    "ca_conv0_ln_gelu_bwd": (
        C.c_int,
        [_vp] * 11 + [_i32, _i64, _i32, _i32, _i32, _f32, _vp],
    ),
    "ca_col2im_1d": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp]),
    "ca_softmax_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i32, _vp]),
    "ca_softmax_bwd": (C.c_int, [_vp, _vp, _vp, _f32, _i32, _i32, _i32, _i64, _vp]),
    "ca_ctc_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "ca_ctc_loss_fwd_bwd": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _vp],
    ),
    "ca_ctc_greedy_decode": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp],
    ),
    "ca_mask_frames": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ca_regroup_pad": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ca_posconv_partial_floats": (_i64, [_i32]),
    "ca_posconv_weight": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ca_posconv_weight_bwd": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    ),
    "ca_cast_f32_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ca_cast_bf16_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ca_transpose_f32_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "ca_conv_weight_reorder": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "ca_conv_weight_grad_reorder": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "ca_sumsq_f32": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "ca_sumsq_ranges_f32": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "ca_clear_ranges": (C.c_int, [_vp, _vp, _i32, _i64, _vp]),
    "ca_sum_f32": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "ca_adamw_step": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _vp, _vp],
    ),
    "ca_adamw_step_ex": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _vp, _i32, _vp],
    ),
    "ca_adamw_step_g16": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _vp, _i32, _vp],
    ),
    "ca_logmel_workspace_bytes": (_i64, [_i32]),
    "ca_logmel": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    "ca_cross_entropy_fwd_bwd": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp],
    ),
    "ca_argmax_masked": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp]),
    "ca_whisper_decode_token": (C.c_int, [C.POINTER(CaDecodeDesc), _vp]),
    "ca_whisper_decode_token_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "ca_debug_decode_stamps": (C.c_int, [_vp, _i32]),
    "ca_argmax_advance": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ca_embed_tokens": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "ca_embed_tokens_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "ca_comm_unique_id": (C.c_int, [_vp]),
    "ca_comm_init": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32]),
    "ca_comm_destroy": (C.c_int, [_vp]),
    "ca_comm_abort": (C.c_int, [_vp]),
    "ca_comm_stream": (_vp, [_vp]),
    "ca_comm_rank": (C.c_int, [_vp]),
    "ca_comm_world": (C.c_int, [_vp]),
    "ca_comm_after": (C.c_int, [_vp, _vp]),
    "ca_comm_before": (C.c_int, [_vp, _vp]),
    "ca_allreduce_bucket": (C.c_int, [_vp, _vp, _i64, _i32]),
    "ca_reduce_scatter_bucket": (C.c_int, [_vp, _vp, _i64, _i32]),
    "ca_allgather_bucket": (C.c_int, [_vp, _vp, _i64, _i32]),
}

_lib = None


def load() -> C.CDLL:
    """Load libcoral_amd.so (once) and attach prototypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CORAL_AMD_LIB", LIB_PATH))
    if not path.exists():
        raise CoralAmdError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C coral_amd/csrc`). There is no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    """Raise CoralAmdError with the library's message if `rc` is not CA_OK."""
    if rc != 0:
        msg = load().ca_last_error()
        raise CoralAmdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
