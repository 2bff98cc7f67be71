"""ctypes binding of ``libcryovit_hip.so`` (the C ABI in ``include/cryovit_hip.h``).

The product path has no fallback: if the library is missing or a call fails, this raises.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# CVX_ABLATION_LIB=1 (tools/ only): the -DCVX_ABLATION build with the timing-only kernel variants (cryovit_amd/build.py)
LIB_PATH = Path(__file__).resolve().parent / ("libcryovit_hip_ablation.so" if os.environ.get("CVX_ABLATION_LIB") == "1"
                                                else "libcryovit_hip.so")

EPI_BF16, EPI_BF16_GELU, EPI_SWIGLU, EPI_RESID, EPI_PATCH, EPI_VT, EPI_CONVT, EPI_F32, EPI_RESID_HL = range(9)
DTYPE_BF16, DTYPE_F16 = 0, 1  # CVX_DTYPE_*
DICE_BLOCKS = 4096  # CVX_DICE_BLOCKS
GN_BLOCKS = 1024  # CVX_GN_BLOCKS
GN_MAX_GROUPS = 512  # CVX_GN_MAX_GROUPS

c_long, c_int, c_float, c_void_p = C.c_long, C.c_int, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("epilogue", c_int),
        ("a", c_void_p), ("lda", c_long),
        ("w", c_void_p), ("ldw", c_long),
        ("m", c_long), ("n", c_long), ("n_pad", c_long), ("k_pad", c_long),
        ("out", c_void_p), ("ldc", c_long),
        ("bias", c_void_p),
        ("gamma", c_void_p),
        ("pos", c_void_p), ("ldpos", c_long),
        ("npatch", c_int), ("ntp", c_int), ("tok0", c_int),
        ("heads", c_int), ("kp", c_int),
        ("H", c_int), ("W", c_int), ("cout", c_int), ("act", c_int), ("dtype", c_int), ("convt_up_z", c_int),
        ("ln_rowstat", c_void_p),
        ("out2", c_void_p), ("stat_part", c_void_p), ("stat_rows", c_long),
    ]


class Conv3dDesc(C.Structure):
    _fields_ = [
        ("in_", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("zero_page", c_void_p), ("out", c_void_p),
        ("C", c_int), ("D", c_int), ("H", c_int), ("W", c_int), ("dil", c_int), ("cout", c_int),
        ("n_pad", c_int), ("k_pad", c_int), ("act", c_int),
    ]


class VitLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "qk_w", "qk_b", "v_w", "v_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b",
                                        "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b", "ls2")]


class VitDesc(C.Structure):
    _fields_ = [("dim", c_int), ("depth", c_int), ("heads", c_int), ("n_reg", c_int), ("ffn_swiglu", c_int), ("hid_pad", c_int),
                ("ln_eps", c_float), ("qkv_merged", c_int), ("ln_fold", c_int), ("pe_b", c_void_p), ("reg", c_void_p), ("norm_w", c_void_p), ("norm_b", c_void_p),
                ("layers", C.POINTER(VitLayer))]


class VitWs(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "xn", "qk", "vt", "ao", "hid", "xh", "xl", "stat_part", "rowstat")]


class HeadBlock(C.Structure):
    _fields_ = [("c1", c_int), ("c2", c_int), ("c3", c_int), ("d1", c_int), ("d2", c_int), ("groups", c_int),
                ("gn_w", c_void_p), ("gn_b", c_void_p),
                ("conv1_w", c_void_p), ("conv1_b", c_void_p), ("conv1_npad", c_int), ("conv1_kpad", c_int),
                ("conv2_w", c_void_p), ("conv2_b", c_void_p), ("conv2_npad", c_int), ("conv2_kpad", c_int),
                ("convt_w", c_void_p), ("convt_b", c_void_p), ("convt_npad", c_int), ("convt_kpad", c_int)]


class HeadDesc(C.Structure):
    _fields_ = [("c_in", c_int), ("c0", c_int), ("c_tail", c_int), ("n_blocks", c_int),
                ("proj_w", c_void_p), ("proj_b", c_void_p), ("proj_npad", c_int), ("proj_kpad", c_int),
                ("blocks", C.POINTER(HeadBlock)),
                ("out0_w", c_void_p), ("out0_b", c_void_p), ("out0_npad", c_int), ("out0_kpad", c_int),
                ("out2_w", c_void_p), ("out2_b", c_float), ("zero_page", c_void_p)]


HEAD_MAX_BLOCKS = 8


class HeadWs(C.Structure):
    _fields_ = [("act0", c_void_p), ("gn", c_void_p * HEAD_MAX_BLOCKS), ("t1", c_void_p * HEAD_MAX_BLOCKS),
                ("t2", c_void_p * HEAD_MAX_BLOCKS), ("up", c_void_p * HEAD_MAX_BLOCKS), ("mid", c_void_p), ("gn_stats", c_void_p),
                ("dice_scratch", c_void_p)]


# name -> (restype, argtypes); every symbol include/cryovit_hip.h declares
SIGNATURES = {
    "cvx_last_error": (C.c_char_p, []),
    "cvx_version": (c_int, []),
    "cvx_device_arch": (c_int, [C.c_char_p, c_int]),
    "cvx_set_option": (c_int, [C.c_char_p, c_int]),
    "cvx_debug_read_gemm256": (c_int, [c_void_p]),
    "cvx_debug_read_gemm256p": (c_int, [c_void_p]),
    "cvx_gemm_bf16": (c_int, [C.POINTER(GemmDesc), c_void_p]),
    "cvx_set_gemm_event_hook": (c_int, [c_int, c_void_p, c_void_p, c_int]),
    "cvx_get_gemm_event_count": (c_int, []),
    "cvx_conv3d_f16": (c_int, [C.POINTER(Conv3dDesc), c_void_p]),
    "cvx_conv2s2_f16": (c_int, [C.POINTER(Conv3dDesc), c_void_p]),
    "cvx_concat_channels_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "cvx_pointwise_out_f16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "cvx_groupnorm_act_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    "cvx_groupnorm_act_strided_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_int,
                                              c_void_p]),
    "cvx_layernorm_bf16": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_float, c_void_p]),
    "cvx_attention_bf16": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cvx_attention_qkv_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "cvx_preprocess_patches": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "cvx_init_tokens": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cvx_final_norm_features": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p]),
    "cvx_final_norm_features_hl": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p]),
    "cvx_split_stream": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_void_p]),
    "cvx_rowstat_finalize": (c_int, [c_void_p, c_int, c_long, c_void_p, c_long, c_int, c_float, c_void_p]),
    "cvx_merge_stream": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "cvx_im2col_patches": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "cvx_features_to_channels_last": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "cvx_groupnorm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_void_p]),
    "cvx_conv3_out_fused": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                    c_int, c_int, c_int, c_void_p]),
    "cvx_sam_patches": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p]),
    "cvx_window_attention_bf16": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_void_p]),
    "cvx_pool2x2": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "cvx_cast_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "cvx_fpn_level_out": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cvx_dice_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p]),
    "cvx_dice_loss_forward": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "cvx_dice_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "cvx_focal_loss_forward": (c_int, [c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "cvx_focal_loss_backward": (c_int, [c_void_p, c_void_p, c_long, c_float, c_void_p, c_float, c_void_p, c_void_p]),
    "cvx_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_double, c_int, c_void_p]),
    "cvx_head_forward": (c_int, [C.POINTER(HeadDesc), C.POINTER(HeadWs), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_float, c_void_p]),
    "cvx_vit_encode": (c_int, [C.POINTER(VitDesc), C.POINTER(VitWs), c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p]),
}

_lib = None


class CvxError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once) and bind every declared symbol; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be in the process BEFORE our library is
    # dlopen()ed so that both bind to ONE HIP runtime; the other order gives two runtimes and "no ROCm-capable device".
    import torch  # noqa: F401

    if not LIB_PATH.exists():
        raise CvxError(
            f"{LIB_PATH} not found: build it with `python -m cryovit_amd.build` (hipcc, gfx950). "
            "cryovit_amd has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().cvx_last_error().decode(errors="replace")
        raise CvxError(f"{what} failed ({rc}): {msg}")


def device_arch() -> str:
    buf = C.create_string_buffer(128)
    check(load().cvx_device_arch(buf, 128), "cvx_device_arch")
    return buf.value.decode()


def set_option(name: str, value: int) -> None:
    check(load().cvx_set_option(name.encode(), int(value)), "cvx_set_option")
