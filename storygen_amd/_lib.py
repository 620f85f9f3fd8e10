"""ctypes binding of libstorygen_hip.so (include/storygen_hip.h).  Fails loudly when the library is missing:
there is no CPU / eager-PyTorch fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstorygen_hip.so")

c_half_p = C.c_void_p   # device pointers travel as integers


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("C2", C.c_void_p), ("ldc2", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("epilogue", C.c_int32),
        ("flags", C.c_int32),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p),
        ("rowbias_ld", C.c_int64),
        ("rows_per_batch", C.c_int32),
        ("split_k", C.c_int32),
        ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("tile_waves", C.c_int32),
        ("res1", C.c_void_p), ("ldr1", C.c_int64),
        ("res2", C.c_void_p), ("ldr2", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("stats", C.c_void_p), ("stats_batch_rows", C.c_int32),
        ("ln_mode", C.c_int32), ("ln_parts", C.c_int32), ("ln_eps", C.c_float),
        ("ln_stats", C.c_void_p), ("ln_c", C.c_void_p), ("ln_d", C.c_void_p), ("ln_stats_out", C.c_void_p),
        ("ln_guard", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("w", C.c_void_p),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("flags", C.c_int32),
        ("stride", C.c_int32),
        ("upsample2x", C.c_int32),
        ("x_padded", C.c_int32),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("rowbias_ld", C.c_int64),
        ("res1", C.c_void_p), ("ldr1", C.c_int64),
        ("split_k", C.c_int32),
        ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("tile_waves", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("stats", C.c_void_p),
        ("defer_reduce", C.c_int32),
    ]


class GroupNormDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("x_f32", C.c_int32),
        ("y_pad_w", C.c_int32),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("xcopy", C.c_void_p), ("ldxc", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("B", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("groups", C.c_int32),
        ("eps", C.c_float),
        ("silu", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("pstats", C.c_void_p * 2),
        ("pstats_rows", C.c_int32 * 2), ("pstats_c0", C.c_int32 * 2), ("pstats_nc", C.c_int32 * 2),
        ("split_ws", C.c_void_p), ("split_count", C.c_int32),
        ("split_bias", C.c_void_p),
        ("split_rowbias", C.c_void_p), ("split_rowbias_ld", C.c_int64),
        ("split_res", C.c_void_p), ("split_ldr", C.c_int64), ("split_res_f32", C.c_int32),
        ("split_out", C.c_void_p), ("split_ldo", C.c_int64), ("split_out_f32", C.c_int32),
        ("split_round_f16", C.c_int32),
    ]


class GroupNormBwdDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64), ("x_f32", C.c_int32),
        ("dy", C.c_void_p), ("lddy", C.c_int64), ("dy_f32", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("res", C.c_void_p), ("ldr", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_f32", C.c_int32), ("out_pad_w", C.c_int32),
        ("B", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("groups", C.c_int32),
        ("eps", C.c_float),
        ("silu", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int64), ("bsq", C.c_int64),
        ("k", C.c_void_p), ("ldk", C.c_int64), ("bsk", C.c_int64),
        ("vt", C.c_void_p), ("ldvt", C.c_int64), ("bsvt", C.c_int64),
        ("o", C.c_void_p), ("ldo", C.c_int64), ("bso", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("D", C.c_int32),
        ("kv_batches", C.c_int32),
        ("scale", C.c_float),
        ("k2", C.c_void_p), ("bsk2", C.c_int64),
        ("vt2", C.c_void_p), ("bsvt2", C.c_int64),
        ("Nk2", C.c_int32), ("kv2_batches", C.c_int32),
    ]


class FfDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("wpack", C.c_void_p), ("wpack_bytes", C.c_size_t),
        ("b2", C.c_void_p),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("M", C.c_int32), ("C", C.c_int32),
        ("eps", C.c_float),
        ("hidden_split", C.c_int32),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int64), ("bsq", C.c_int64),
        ("qt", C.c_void_p), ("ldqt", C.c_int64), ("bsqt", C.c_int64),
        ("k", C.c_void_p), ("ldk", C.c_int64), ("bsk", C.c_int64),
        ("kt", C.c_void_p), ("ldkt", C.c_int64), ("bskt", C.c_int64),
        ("v", C.c_void_p), ("ldv", C.c_int64), ("bsv", C.c_int64),
        ("dout", C.c_void_p), ("lddo", C.c_int64), ("bsdo", C.c_int64),
        ("dot", C.c_void_p), ("lddot", C.c_int64), ("bsdot", C.c_int64),
        ("ld2", C.c_void_p),
        ("dq", C.c_void_p), ("lddq", C.c_int64), ("bsdq", C.c_int64),
        ("dkt", C.c_void_p), ("lddkt", C.c_int64), ("bsdkt", C.c_int64),
        ("dvt", C.c_void_p), ("lddvt", C.c_int64), ("bsdvt", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("D", C.c_int32),
        ("scale", C.c_float),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header.
class AdamWDesc(C.Structure):
    _fields_ = [
        ("param", C.c_void_p), ("grad", C.c_void_p), ("n", C.c_int64),
        ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("code1", C.c_void_p), ("code2", C.c_void_p),
        ("absmax1", C.c_void_p), ("absmax2", C.c_void_p),
        ("q_code1", C.c_void_p), ("q_code2", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("step", C.c_int32),
        ("grad_scale", C.c_float),
        ("sumsq", C.c_void_p), ("n_sumsq", C.c_int32), ("max_norm", C.c_float),
    ]


SIGNATURES = {
    "sg_version": (C.c_int, []),
    "sg_last_error": (C.c_char_p, []),
    "sg_device_arch": (C.c_int, []),
    "sg_device_cus": (C.c_int, []),
    "sg_gemm_f16": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "sg_gemm_stats_tile_rows": (C.c_int, [C.POINTER(GemmDesc)]),
    "sg_conv3x3_stats_tile_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "sg_conv3x3_planned_splits": (C.c_int, [C.POINTER(ConvDesc)]),
    "sg_gemm_launch_plan": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(C.c_int32)]),
    "sg_conv3x3_launch_plan": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    "sg_groupnorm_uses_pstats": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "sg_groupnorm_is_fused": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "sg_ff_fused_pack_bytes": (C.c_size_t, [C.c_int32]),
    "sg_ff_geglu_fused_f16": (C.c_int, [C.POINTER(FfDesc), C.c_void_p]),
    "sg_gemm_pair_f16": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), C.c_void_p]),
    "sg_gemm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sg_conv3x3_nhwc_f16": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "sg_conv_in_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_conv_out_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_attn_fwd_f16": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p]),
    "sg_attn_fwd_pair_f16": (C.c_int, [C.POINTER(AttnDesc), C.POINTER(AttnDesc), C.c_void_p]),
    "sg_groupnorm_nhwc_f16": (C.c_int, [C.POINTER(GroupNormDesc), C.c_void_p]),
    "sg_groupnorm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sg_layernorm_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sg_timestep_embed_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_lookup_rows_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "sg_linear_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_add_noise_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                   C.c_void_p]),
    "sg_cfg_ddim_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "sg_cfg_plms_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                       C.c_void_p]),
    "sg_copy_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_void_p]),
    "sg_pad_cast_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "sg_attn_f8_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sg_attn_f8_pack": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_attn_fwd_f8_d40": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "sg_softmax_rows_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "sg_attn_small_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                    C.c_int32, C.c_void_p]),
    "sg_act_rows_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_embed_tokens_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p]),
    "sg_gaussian_sample_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "sg_sumsq_scratch_floats": (C.c_size_t, []),
    "sg_sumsq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sg_adamw_f32": (C.c_int, [C.POINTER(AdamWDesc), C.c_void_p]),
    "sg_adamw8bit_blocks": (C.c_size_t, [C.c_int64]),
    "sg_adamw8bit": (C.c_int, [C.POINTER(AdamWDesc), C.c_void_p]),
    "sg_attn_fwd_lse_f16": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p, C.c_void_p]),
    "sg_attn_bwd_prep_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_attn_bwd_dq_f16": (C.c_int, [C.POINTER(AttnBwdDesc), C.c_void_p]),
    "sg_attn_bwd_dkv_f16": (C.c_int, [C.POINTER(AttnBwdDesc), C.c_void_p]),
    "sg_layernorm_bwd_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_int32, C.c_float, C.c_void_p]),
    "sg_geglu_bwd_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_void_p]),
    "sg_groupnorm_bwd_nhwc_f16": (C.c_int, [C.POINTER(GroupNormBwdDesc), C.c_void_p]),
    "sg_groupnorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sg_transpose_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_transpose_batched_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p]),
    "sg_sum2x2_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "sg_zero_stuff_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "sg_mse_grad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sg_debug_mfma_32x32x16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sg_debug_mfma_f8_32x32x64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sg_debug_set_tile": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "sg_debug_fastdiv_selftest": (C.c_int, []),
    "sg_debug_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "sg_debug_gemm_anatomy": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "sg_debug_conv_anatomy": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "sg_debug_ff_anatomy": (C.c_int, [C.POINTER(FfDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the library and bind every declared symbol; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m storygen_amd.build` (hipcc, gfx950). "
            "storygen_amd has no CPU / eager fallback.")
    # libstorygen_hip.so and PyTorch-ROCm must share ONE HIP runtime (same soname libamdhip64.so.7): import torch
    # first so that the dynamic loader resolves our dependency to the copy torch already mapped.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().sg_last_error().decode(errors="replace")
        raise RuntimeError(f"libstorygen_hip {what} failed ({code}): {msg}")
