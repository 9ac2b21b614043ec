"""ctypes binding of libstedm_hip.so (the C ABI declared in include/stedm_hip.h).

The library is the product: there is NO fallback. If it is missing or a call fails, an
exception is raised (`StedmHipError`)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEDM_HIP_LIB") or os.path.join(_HERE, "libstedm_hip.so")     # STEDM_HIP_LIB: A/B timing of another build

ABI_VERSION = 12
F16, BF16 = 0, 1
CONV_S1, CONV_DOWN, CONV_UP, CONV_UP_SUBPIXEL, CONV_S2D = 0, 1, 2, 3, 4


class StedmHipError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    """struct stedm_conv_args (include/stedm_hip.h)."""
    _fields_ = [
        ("src1", C.c_void_p), ("src2", C.c_void_p),
        ("c1", C.c_int32), ("c2", C.c_int32), ("src2_bmod", C.c_int32),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("mode", C.c_int32), ("ks", C.c_int32),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("act", C.c_int32),
        ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("bias", C.c_void_p),
        ("emb", C.c_void_p), ("emb_bstride", C.c_int32),
        ("res", C.c_void_p), ("out", C.c_void_p),
        ("cout", C.c_int32), ("npass", C.c_int32), ("mm_dtype", C.c_int32),
        ("src16_hi", C.c_void_p), ("src16_lo", C.c_void_p),
        ("act_out", C.c_int32), ("out16_hi", C.c_void_p), ("out16_lo", C.c_void_p),
        ("w_frag", C.c_void_p), ("chan_stats", C.c_void_p),
        ("src16b_hi", C.c_void_p), ("w_frag_b", C.c_void_p), ("bias_b", C.c_void_p), ("cb", C.c_int32),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("chan_nslab", C.c_int32), ("w_frag16", C.c_void_p), ("w_frag_b16", C.c_void_p), ("pad_br", C.c_int32),
        ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_eps", C.c_float), ("gn_groups", C.c_int32), ("gn_act", C.c_int32), ("gn_out16", C.c_void_p), ("gn_mr", C.c_void_p), ("gn_only", C.c_int32),
        ("gn_out16_lo", C.c_void_p),
        ("qkv_q", C.c_void_p), ("qkv_k", C.c_void_p), ("qkv_vt", C.c_void_p),
        ("qkv_T", C.c_int32), ("qkv_Tp", C.c_int32), ("qkv_heads", C.c_int32), ("qkv_qscale", C.c_float),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_res", C.c_void_p), ("ln_eps", C.c_float),
        ("out16_stride", C.c_int32),
        ("gn_coop", C.c_void_p), ("gn_coop_epoch", C.c_void_p), ("gn_coop_tmo", C.c_void_p),
    ]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
# symbol -> (restype, argtypes); must list EVERY symbol of include/stedm_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "stedm_abi_version": (_I, []),
    "stedm_last_error": (C.c_char_p, []),
    "stedm_device_cus": (_I, []),
    "stedm_f16_guard_set": (_I, [_P]),
    "stedm_pack_conv_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_up": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_frag": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_up_frag": (_I, [_P, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_frag16": (_I, [_P, C.c_long, C.c_long, _I, _P, _I, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_frag16_hl": (_I, [_P, C.c_long, C.c_long, _I, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_frag16_hl1": (_I, [_P, C.c_long, C.c_long, _I, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_up_frag16_hl": (_I, [_P, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_s2d_frag16_hl": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_s2d_frag": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_space_to_depth16": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "stedm_transpose_f32": (_I, [_P, _P, _I, _I, _P]),
    "stedm_gn_scale_shift": (_I, [_P, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _P, _P, _P]),
    "stedm_gn_nslab": (_I, [_I, _I]),
    "stedm_gn_stats": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "stedm_gn_apply16": (_I, [_P, _I, _P, _I, _I, _P, _P, _F, _I, _I, _P, _I, _I, _P, _P, _I, _P]),
    "stedm_gn_chan_nslab": (_I, [_I]),
    "stedm_gn_chan_stats": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "stedm_gn_chan_stats16": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "stedm_gn_apply16c": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "stedm_gn_apply16c_mr": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "stedm_conv3x3_tiles_ok": (_I, [_I, _I]),
    "stedm_im2col_rows16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_philox_normal": (_I, [_P, _I, _I, _P, _I, C.c_ulonglong, C.c_uint, _P]),
    "stedm_gn_apply16c_x16": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _I, _P, _P, _I, _P]),
    "stedm_conv_igemm": (_I, [C.POINTER(ConvArgs), _P]),
    "stedm_conv_fused_skip_ok": (_I, [C.POINTER(ConvArgs)]),
    "stedm_conv_rs_ok": (_I, [C.POINTER(ConvArgs)]),
    "stedm_conv_in": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "stedm_conv_out": (_I, [_P, _I, _P, _I, _P, _P, _F, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "stedm_time_embed": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "stedm_emb_proj": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "stedm_linear": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_attn_legacy": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_attn_legacy16": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_ddim_step": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _I, _I, _I, _I, _P]),
    "stedm_step_advance": (_I, [_P, _I, _P]),
    "stedm_step_set_t": (_I, [_P, _P, _P, _I, _P]),
    "stedm_svit_patch_embed": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _I, _P]),
    "stedm_svit_patch_ln16": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _I, _P]),
    "stedm_svit_tok_place": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "stedm_ln_apply16": (_I, [_P, _P, _P, _F, _P, _P, C.c_long, _I, _I, _P]),
    "stedm_qkv_pack": (_I, [_P, _I, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_lsa_flash": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_lsa_flash_drop": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, C.c_ulonglong, C.c_uint, _P]),
    "stedm_dropout_rows": (_I, [_P, _P, _P, _P, _P, C.c_long, _F, C.c_ulonglong, C.c_uint, _I, _P]),
    "stedm_qkv_pack_mx8": (_I, [_P, _I, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_lsa_flash_mx8": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_svit_head": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _I, _P, C.c_long, _P]),
    "stedm_ln_bwd_blocks": (_I, [C.c_long]),
    "stedm_ln_bwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, C.c_long, _I, _I, _P]),
    "stedm_geglu_bwd": (_I, [_P, _P, _P, C.c_long, _I, _P]),
    "stedm_geglu16": (_I, [_P, _P, _P, C.c_long, _I, _I, _P]),
    "stedm_agg_reduce": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_spatial_rescale": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_pack_frag_multi": (_I, [_P, _I, _I, _I, _P]),
    "stedm_swin_patch16": (_I, [_P, C.c_long, C.c_long, C.c_long, C.c_long, _I, _I, _I, _P, _P, _I, _P]),
    "stedm_swin_ln": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, C.c_long, _I, _I, _I, _P]),
    "stedm_swin_ln_gated": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, C.c_long, _I, _I, _P, _I, _I, _P]),
    "stedm_swin_window_attn": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_swin_merge16": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "stedm_swin_rpb": (_I, [_P, _P, _P, _I, _I, _P]),
    "stedm_swin_token_mean": (_I, [_P, _P, _I, _I, _I, _P]),
    "stedm_pack_conv_weight_strided": (_I, [_P, C.c_long, C.c_long, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "stedm_gn_fold": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "stedm_gn_bwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P]),
    "stedm_gn_bwd_ws_floats": (C.c_long, [_I, _I, _I, _I]),
    "stedm_im2col_t16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.c_long, _P]),
    "stedm_wgrad_to_oihw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_wgrad3x3_plan": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_int)]),
    "stedm_wgrad3x3": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_wgrad3x3_oihw": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_sum_planes": (_I, [_P, _P, C.c_long, _I, _I, _P]),
    "stedm_wgrad1x1_plan": (_I, [C.c_long, _I, _I, C.POINTER(C.c_int)]),
    "stedm_wgrad1x1": (_I, [_P, _P, _P, C.c_long, _I, _I, _I, _P]),
    "stedm_chan_sum_fold": (_I, [_P, _I, _I, _I, _P, C.c_long, _P, _I, _P]),
    "stedm_chan_sum_fold2": (_I, [_P, _I, _I, _I, _P, C.c_long, _P, _I, _P, _P]),
    "stedm_sum2x2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_zero_insert16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "stedm_attn_legacy_bwd_ws_floats": (C.c_long, [_I, _I, _I]),
    "stedm_attn_legacy_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "stedm_gemm_f32": (_I, [_P, C.c_long, _I, _P, C.c_long, _I, _P, C.c_long, _I, _I, _I, _F, _F, _P, C.c_long, _P]),
    "stedm_silu": (_I, [_P, _P, _P, C.c_long, _I, _P]),
    "stedm_q_sample": (_I, [_P, _P, _P, _P, _P, _P, _I, C.c_long, _P]),
    "stedm_l1_loss": (_I, [_P, _P, C.c_long, _F, _P, _P, _P, _P]),
    "stedm_spatial_rescale_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "stedm_axpby_f32": (_I, [_P, _P, C.c_long, _F, _F, _P]),
    "stedm_adamw_ema": (_I, [_P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _F, _F, _P]),
    "stedm_adamw_ema_pack": (_I, [_P, _I, _I, _F, _F, _F, _F, _F, _I, _F, _F, _P]),
    "stedm_adamw_ema_pack_piece": (_I, [_P, _P]),
    "stedm_adamw_ema_sched": (_I, [_P, _P, _P, _I, _F, _F, _F, _F, _P, _P, _F, _P]),
    "stedm_adamw_ema_pack_sched": (_I, [_P, _I, _I, _F, _F, _F, _F, _P, _P, _F, _P]),
    "stedm_ema_update": (_I, [_P, _P, _P, _I, _F, _P]),
    "stedm_vq_nearest": (_I, [_P, _P, _I, _I, _I, C.c_long, _P, _P, _P]),
    "stedm_conv1x1_nchw": (_I, [_P, _P, _P, _P, _I, _I, _I, C.c_long, _P]),
    "stedm_softmax_rows16": (_I, [_P, C.c_long, _F, _P, _P, C.c_long, _I, C.c_long, _I, _P]),
    "stedm_image_to_uint8": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_seg_merge": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stedm_argmax_u8": (_I, [_P, _P, C.c_long, _I, _P]),
    "stedm_graph_begin": (_I, [_P]),
    "stedm_graph_end": (_I, [_P, C.POINTER(C.c_void_p)]),
    "stedm_graph_launch": (_I, [_P, _P]),
    "stedm_graph_destroy": (_I, [_P]),
    "stedm_debug_conv_stamps": (_I, [_P, _I]),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises StedmHipError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StedmHipError(
                f"{LIB_PATH} not found: build it with `python -m stedm_amd.build` "
                "(there is no CPU / PyTorch fallback for the hot path)")
        try:
            # One HIP runtime per process: the PyTorch-ROCm wheel bundles libamdhip64.so (soname .so.7) under
            # torch/lib. Loaded first, it satisfies this library's NEEDED libamdhip64.so.7 by soname; loaded second,
            # /opt/rocm's copy would already be in and torch would add its own -> two runtimes, "no device".
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise StedmHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.stedm_abi_version() != ABI_VERSION:
            raise StedmHipError(f"ABI mismatch: library {L.stedm_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().stedm_last_error().decode(errors="replace")
        raise StedmHipError(f"{what} failed (rc={rc}): {msg}")
