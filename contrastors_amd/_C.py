"""ctypes binding of libcontrastors_hip.so (the C-ABI declared in include/contrastors_hip.h).

There is NO fallback: if the shared object is missing or a symbol is absent, importing/using this module raises.
The product path never routes through oracle/ or any CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CONTRASTORS_HIP_LIB", _PKG / "lib" / "libcontrastors_hip.so"))

CX_OK = 0
_ERR = {-1: "CX_ERR_SHAPE", -2: "CX_ERR_ALIGN", -3: "CX_ERR_ARG", -4: "CX_ERR_LAUNCH"}

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_long
f32 = C.c_float
u64 = C.c_ulonglong
u32 = C.c_uint


class CxLayerWeights(C.Structure):
    _fields_ = (
        [(n, vp) for n in ("Wqkv", "Wout", "Wfc1", "Wfc2", "WqkvT", "WoutT", "Wfc1T", "Wfc2T")]
        + [(n, vp) for n in ("bqkv", "bout", "bfc1", "bfc2")]
        + [(n, vp) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b")]
        + [(n, vp) for n in ("gWqkv", "gWout", "gWfc1", "gWfc2")]
        + [(n, vp) for n in ("gbqkv", "gbout", "gbfc1", "gbfc2")]
        + [(n, vp) for n in ("gln1_g", "gln1_b", "gln2_g", "gln2_b")]
    )


class CxEncoderDesc(C.Structure):
    _fields_ = [
        ("n_layer", i32), ("d", i32), ("n_head", i32), ("d_inner", i32), ("gated", i32),
        ("vocab", i32), ("max_pos", i32), ("padding_idx", i32),
        ("ln_eps", f32), ("softmax_scale", f32),
        ("word_emb", vp), ("type_emb", vp), ("pos_emb", vp),
        ("emb_ln_g", vp), ("emb_ln_b", vp),
        ("gword_emb", vp), ("gtype_emb", vp), ("gpos_emb", vp), ("gemb_ln_g", vp), ("gemb_ln_b", vp),
        ("rot_cos", vp), ("rot_sin", vp),
        ("layers", C.POINTER(CxLayerWeights)),
        ("pool_mode", i32), ("normalize", i32),
        ("prenorm", i32),
        ("lnf_g", vp), ("lnf_b", vp), ("glnf_g", vp), ("glnf_b", vp),
        ("Wpatch", vp), ("bpatch", vp), ("cls_token", vp), ("vit_pos", vp),
        ("gWpatch", vp), ("gbpatch", vp), ("gcls_token", vp), ("gvit_pos", vp),
        ("patch_dim", i32),
        ("resid_pdrop", f32), ("embd_pdrop", f32), ("attn_pdrop", f32),
        ("mlp_act", i32), ("lnpre_g", vp), ("lnpre_b", vp), ("glnpre_g", vp), ("glnpre_b", vp),
    ]


class CxChunkBuffers(C.Structure):
    _fields_ = [("T_cap", i64)] + [
        (n, vp)
        for n in (
            "h0", "emb_mean", "emb_rstd", "qkv", "ctx", "lse", "z1", "h1", "mean1", "rstd1", "yg", "act", "z2",
            "h2", "mean2", "rstd2", "pool_norm", "g_a", "g_b", "g_c", "g_wide", "g_act", "tr_a", "tr_b", "delta",
            "ws_f32",
        )
    ] + [("ws_floats", i64)] + [(n, vp) for n in ("zf", "hf", "meanf", "rstdf", "patch_in", "patch_proj")] + [("checkpoint", i32), ("drop_active", i32), ("drop_seed", C.c_ulonglong), ("drop_offset", C.c_ulonglong), ("g_d", vp), ("layer_events", C.POINTER(vp)), ("zpre", vp), ("ckpt_keep", i32), ("patch_keep", vp), ("patch_inv", vp), ("n_keep", i32), ("n_patch_all", i32)]


# name -> (restype, argtypes).  Keep in the order of include/contrastors_hip.h.
_SIGS = {
    "cx_abi_version": (i32, []),
    "cx_build_info": (C.c_char_p, []),
    "cx_error_string": (C.c_char_p, [i32]),
    "cx_gemm_bf16_nt": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "cx_gemm_bf16_nt_accum": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_tn_accum": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "cx_prof_gemm_config": (i32, [i32, i32]),
    "cx_prof_gemm_collect": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64)]),
    "cx_calib_mfma_bf16": (i32, [vp, i32, i32, vp, vp, vp]),
    "cx_calib_copy": (i32, [vp, vp, i64, vp]),
    "cx_transpose_bf16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_cast_f32_to_bf16": (i32, [vp, vp, i64, vp]),
    "cx_cast_transpose_f32_to_bf16": (i32, [vp, vp, i32, i32, vp]),
    "cx_cast_transpose_f32_to_bf16_batched": (i32, [vp, i32, i32, vp]),
    "cx_cast_bf16_to_f32": (i32, [vp, vp, i64, vp]),
    "cx_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "cx_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "cx_layernorm_bwd_pooled": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "cx_layernorm_bwd_colsum": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "cx_dropout_add_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, C.c_ulonglong, C.c_ulonglong,
                                           C.c_uint, vp]),
    "cx_dropout_add_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, C.c_ulonglong,
                                           C.c_ulonglong, C.c_uint, vp]),
    "cx_dropout_add_layernorm_bwd_colsum": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, C.c_ulonglong,
                                                  C.c_ulonglong, C.c_uint, vp]),
    "cx_dropout_scale": (i32, [vp, i64, f32, C.c_ulonglong, C.c_ulonglong, C.c_uint, vp]),
    "cx_layernorm_fwd_mixed": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]),
    "cx_layernorm_bwd_mixed": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cx_embed_ln_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "cx_embed_ln_bwd": (i32, [vp] * 15 + [i32, i32, i32, i32, vp]),
    "cx_embed_ln_bwd_sorted": (i32, [vp] * 15 + [i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "cx_swiglu_fwd": (i32, [vp, vp, i32, i32, i32, vp]),
    "cx_swiglu_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cx_gemm_bf16_swiglu": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_nt_residual": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_swiglu_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_nt_splitk": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_swiglu_gate": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_swiglu_bwd_gate": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_swiglu_bwd_gate": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "cx_gemm_bf16_bias_gelu": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_gemm_bf16_bias_act": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_bias_gelu_fwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "cx_bias_gelu_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "cx_bias_grad": (i32, [vp, vp, i32, i32, i32, vp]),
    "cx_bias_gelu_bwd_colsum": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    "cx_bias_act_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cx_bias_act_bwd_colsum": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cx_gemm_bf16_act_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cx_attn_varlen_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "cx_attn_varlen_bwd": (i32, [vp] * 9 + [i32, i32, i32, i32, f32, vp]),
    "cx_attn_varlen_bwd_prerotated": (i32, [vp] * 9 + [i32, i32, i32, i32, f32, vp]),
    "cx_attn_varlen_dropout_fwd": (i32, [vp] * 6 + [i32, i32, i32, i32, f32, f32, u64, u64, u32, vp]),
    "cx_attn_varlen_dropout_bwd": (i32, [vp] * 9 + [i32, i32, i32, i32, f32, f32, u64, u64, u32, vp]),
    "cx_ipc_alloc": (i32, [C.POINTER(vp), i64, i32]),
    "cx_ipc_free": (i32, [vp]),
    "cx_ipc_export": (i32, [vp, vp]),
    "cx_ipc_open": (i32, [vp, C.POINTER(vp)]),
    "cx_ipc_close": (i32, [vp]),
    "cx_xgmi_push": (i32, [vp, vp, i64, i64, i32, vp]),
    "cx_xgmi_scatter": (i32, [vp, vp, i32, i64, i32, vp]),
    "cx_xgmi_signal_wait": (i32, [vp, vp, i32, i32, u32, i64, vp, vp]),
    "cx_sum_slots_f32": (i32, [vp, vp, i64, i32, vp]),
    "cx_attn_varlen_kvpacked_fwd": (i32, [vp] * 6 + [i32, i32, i32, i32, i32, f32, vp]),
    "cx_attn_varlen_kvpacked_bwd": (i32, [vp] * 10 + [i32, i32, i32, i32, i32, f32, vp]),
    "cx_rotary_qkv_inplace": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_rotary_apply": (i32, [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_pool_normalize_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "cx_pool_normalize_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "cx_vit_patchify": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "cx_vit_assemble_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "cx_vit_assemble_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "cx_vit_patchify_gather": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "cx_vit_assemble_fwd_gather": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "cx_vit_assemble_bwd_gather": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, i32, vp]),
    "cx_vit_forward": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, i32, vp, i32, i32, i32, i32, i32,
                             i32, vp, vp]),
    "cx_vit_backward": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, i32, i32, vp, vp, vp]),
    "cx_vit_forward_hidden": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, i32, vp, i32, i32, i32, i32, i32,
                                    i32, vp, vp]),
    "cx_vit_backward_hidden": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, i32, i32, vp, vp]),
    "cx_xent_fwd": (i32, [vp, i32, vp, vp, vp, i32, i32, i64, f32, i64, vp]),
    "cx_xent_bwd": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i64, i64, f32, i64, vp]),
    "cx_grad_sq_norm": (i32, [vp, i64, vp, vp]),
    "cx_adamw_clip_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp, f32, vp]),
    "cx_ema_update": (i32, [vp, vp, i64, f32, vp]),
    "cx_infonce_ws_floats": (i64, [i32, i32]),
    "cx_infonce_fwd": (i32, [vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_infonce_argmax_ws_floats": (i64, [i32, i32]),
    "cx_infonce_fwd_argmax": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_infonce_bwd": (i32, [vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_infonce_fp8_ws_floats": (i64, [i32, i32]),
    "cx_infonce_fp8_fwd": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cx_infonce_fp8_bwd": (i32, [vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32,
                                 i32, i32, vp]),
    "cx_sgemm_nt": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "cx_transpose_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "cx_encoder_forward": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, vp, vp, i32, i32, i32,
                                 i32, i32, vp, vp]),
    "cx_encoder_forward_hidden": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, vp, vp, i32, i32, i32,
                                        i32, i32, vp, vp]),
    "cx_encoder_backward_hidden": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, vp, vp, i32, i32, i32,
                                         i32, vp, vp, vp, vp]),
    "cx_encoder_backward": (i32, [C.POINTER(CxEncoderDesc), C.POINTER(CxChunkBuffers), vp, vp, vp, i32, i32, i32,
                                  i32, vp, vp, vp, vp, vp]),
}

# development-only entry points (include/contrastors_hip_dev.h): exported by libcontrastors_hip_dev.so, which is the same
# code built without -DCX_PRODUCT (+ the earlier GEMM generations, A/B attention kernels, probes)
_DEV_SIGS = {
    "cx_gemm_set_variant": (None, [i32]),
    "cx_gemm_get_variant": (i32, []),
    "cx_gemm_set_debug": (None, [i32]),
    "cx_gemm_set_trace": (None, [vp]),
    "cx_gemm_v6_ablate": (None, [i32]),
    "cx_gemm_v6_trace": (None, [vp]),
    "cx_gemm_v7_mode": (None, [i32]),
    "cx_gemm_v7_trace": (None, [vp]),
    "cx_gemm_v7_occupancy": (i32, []),
    "cx_gemm_v7_ablate": (None, [i32]),
    "cx_gemm_v7_flags": (None, [i32]),
    "cx_gemm_v7_period": (None, [i32]),
    "cx_gemm_set_glds": (None, [i32]),
    "cx_gemm_get_glds": (i32, []),
    "cx_attn_set_bwd_s128": (None, [i32]),
    "cx_attn_set_prio": (None, [i32]),
    "cx_attn_set_fwd_s128": (None, [i32]),
    "cx_attn_set_fwd_long": (None, [i32]),
    "cx_attn_set_bwd_long": (None, [i32]),
    "cx_probe_mfma_layout": (i32, [vp, vp]),
    "cx_probe_ds_read_tr16": (i32, [vp, vp, vp]),
    "cx_probe_mfma_rate": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "cx_attn_dropout_keep_mask": (i32, [vp, i32, i32, i32, f32, u64, u64, u32, vp]),
    "cx_probe_rmw": (i32, [vp, i64, i32, i32, i32, vp]),
    "cx_probe_mfma_rate16": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "cx_probe_dma_bw": (i32, [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
DEV_EXPORTED_SYMBOLS = tuple(_DEV_SIGS)
DEV_LIB_PATH = Path(os.environ.get("CONTRASTORS_HIP_DEV_LIB", _PKG / "lib" / "libcontrastors_hip_dev.so"))

_lib = None
_dev_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library with typed entry points.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m contrastors_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # libcontrastors_hip.so needs libamdhip64.so.7; torch bundles its own copy with the same SONAME.  Import torch
        # FIRST so that copy is the single HIP runtime of the process (two runtimes = launches on foreign streams fail).
        import torch  # noqa: F401

        h = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def dev_lib() -> C.CDLL:
    """The development library (scripts/, A/B parity tests): product entry points + the switches of _DEV_SIGS."""
    global _dev_lib
    if _dev_lib is None:
        if not DEV_LIB_PATH.exists():
            raise ImportError(f"{DEV_LIB_PATH} not found: build it with `python -m contrastors_amd.build`")
        import torch  # noqa: F401

        h = C.CDLL(str(DEV_LIB_PATH))
        for name, (res, args) in {**_SIGS, **_DEV_SIGS}.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _dev_lib = h
    return _dev_lib


def check(rc: int, what: str = "") -> None:
    if rc != CX_OK:
        raise RuntimeError(f"contrastors_hip: {what or 'call'} failed with {_ERR.get(rc, rc)}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def cur_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
