"""ctypes binding of include/diffsensei_hip.h.

The library is the product: if it cannot be loaded this module raises — there is no PyTorch/CPU fallback
anywhere in `diffsensei_amd` (a silent fallback would void every parity and performance claim).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFSENSEI_LIB: load another build of the SAME library (A/B of compiler flags, tools/gpu_*): never a fallback - it must exist
LIB_PATH = os.environ.get("DIFFSENSEI_LIB") or os.path.join(_HERE, "lib", "libdiffsensei_hip.so")

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class DsOp(C.Structure):
    _fields_ = [("code", C.c_int32), ("i", C.c_int32 * 16), ("f", C.c_float * 4), ("l", C.c_int64 * 12),
                ("p", C.c_void_p * 10)]


# opcode values of enum ds_opcode
OP = dict(GEMM=1, CONV3X3=2, GROUPNORM=3, LAYERNORM=4, SELF_ATTN=5, IP_ATTN=6, CONV_IN=7, CONV_OUT=8, SKINNY=9,
          TIMESTEP_EMBED=10, ADD_TIME_IDS=11, SAMPLER_STEP=12, PREP_INPUT=13, ADVANCE=14, NHWC2NCHW=15, NCHW2NHWC=16,
          PAD_ROWS=17, SMALL_ATTN=18, LLM_GEMV=19, LLM_ATTN=20, LLM_RMSNORM=21, LLM_EMBED=22, LLM_SELECT=23,
          LLM_ADVANCE=24, LN_FINALIZE=27)

# name -> (restype, argtypes).  Every symbol declared in include/diffsensei_hip.h appears here;
# tests/test_capi_symbols.py checks the two lists against each other.
SIGNATURES = {
    "ds_last_error": (C.c_char_p, []),
    "ds_version": (i32, []),
    "ds_device_info": (i32, [C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    "ds_set_option": (i32, [C.c_char_p, i32]),
    "ds_debug_counter": (i32, [C.c_char_p, i32, C.POINTER(C.c_longlong)]),
    "ds_gemm_t160_fits": (i32, [i32, i32, i32, i32]),
    "ds_gemm_g320_fits": (i32, [i32, i32, i32, i32]),
    "ds_conv3x3_gn_chunks": (i32, [i32, i32, i32, i32, i32]),
    "ds_gemm_f16": (i32, [vp, i64, vp, i64, i32, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "ds_gemm_ln_f16": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i32, i32, i32, i32, vp]),
    "ds_ln_finalize": (i32, [vp, vp, i32, i32, i32, f32, vp]),
    "ds_gemm_ln_fusable": (i32, [i32, i32, i32, i32, i32]),
    "ds_gemm_ln_partial_f16": (i32, [vp, i64, vp, i64, vp, vp, f32, vp, vp, i64, vp, i64, vp, i32, i32, i32, i32, vp]),
    "ds_gemm_ln_swapped_f16": (i32, [vp, i64, vp, i64, i64, vp, i64, vp, vp, i64, i64, i32, i32, i32, i32, vp]),
    "ds_gemm_ln_swapped_partial_f16": (i32, [vp, i64, vp, i64, i64, vp, f32, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, vp]),
    "ds_gemm_f16_batched": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    "ds_conv3x3_f16": (i32, [vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ds_conv3x3_resize_f16": (i32, [vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ds_conv3x3_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ds_gemm_bf16": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, vp]),
    "ds_gemm_bf16_batched": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    "ds_groupnorm_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "ds_wide_attn_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "ds_vae_conv_in_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ds_vae_conv_out_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ds_groupnorm_scaled_f16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, vp]),
    "ds_wide_attn_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "ds_vae_conv_in_f16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ds_vae_conv_out_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ds_groupnorm_workspace_bytes": (sz, [i32, i32]),
    "ds_groupnorm_f16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]),
    "ds_layernorm_f16": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "ds_self_attn_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, f32, vp]),
    "ds_masked_ip_attn_f16": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                    i32, f32, f32, vp, i64, i64, i64, vp]),
    "ds_ip_region_flags": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ds_small_attn_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, f32,
                                vp]),
    "ds_small_attn_causal_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, f32, vp]),
    "ds_embed_tokens_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ds_conv_in_dialog_f16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ds_conv_out_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ds_skinny_linear_f16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ds_timestep_embed_f16": (i32, [vp, vp, vp, i32, i32, i32, f32, vp]),
    "ds_add_time_ids_f16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "ds_cfg_sampler_step_f16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ds_prepare_model_input_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds_nhwc_to_nchw_f16": (i32, [vp, vp, i32, i32, i32, vp]),
    "ds_nchw_to_nhwc_f16": (i32, [vp, vp, i32, i32, i32, vp]),
    "ds_pad_rows_f16": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ds_image_f32_to_u8_nhwc": (i32, [vp, vp, i32, i32, i32, vp]),
    "ds_llm_gemv_f16": (i32, [vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp, i32, f32, vp]),
    "ds_llm_attn_f16": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, f32, vp]),
    "ds_llm_rmsnorm_f16": (i32, [vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, f32, vp]),
    "ds_llm_embed_f16": (i32, [vp, vp, vp, i32, i32, vp]),
    "ds_llm_select_f16": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, vp]),
    "ds_llm_advance": (i32, [vp, i32, vp]),
    "ds_blend_f16": (i32, [vp, vp, vp, i64, f32, vp]),
    "ds_llm_swiglu_f16": (i32, [vp, vp, i32, i32, vp]),
    "ds_resize_h_u8": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp]),
    "ds_resize_v_norm_u8": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, C.POINTER(f32), C.POINTER(f32),
                                  vp, vp, vp]),
    "ds_op_run": (i32, [C.POINTER(DsOp), vp]),
    "ds_op_describe": (i32, [C.POINTER(DsOp), C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ds_plan_create": (i32, [C.POINTER(DsOp), i32, C.POINTER(vp)]),
    "ds_plan_num_ops": (i32, [vp]),
    "ds_plan_run": (i32, [vp, vp]),
    "ds_plan_capture": (i32, [vp, vp]),
    "ds_plan_replay": (i32, [vp, vp]),
    "ds_plan_destroy": (i32, [vp]),
}

_lib = None
_tls = threading.local()   # which threads have had the environment's A/B options applied


class DiffSenseiHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libdiffsensei_hip.so (built by `python -m diffsensei_amd.build`); raise loudly when absent."""
    global _lib
    if _lib is not None:
        if not getattr(_tls, "env_applied", False):
            apply_env_options(_lib)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffSenseiHipError(
            f"{LIB_PATH} is missing: the HIP kernel library is the only execution path of diffsensei_amd. "
            f"Build it with `python -m diffsensei_amd.build` (hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    apply_env_options(lib)
    return lib


def apply_env_options(lib=None) -> None:
    """Apply DS_OPTIONS / DS_GEMM_VARIANT (A/B knobs from the environment) to the CALLING thread.

    The library's knobs are thread-local (include/diffsensei_hip.h): a value set with ds_set_option changes only the launches
    - and the host-side plan queries such as ds_gemm_ln_fusable - of the thread that set it.  The environment is a process-wide
    request, so `load()` applies it once per thread, on that thread's first call: a plan built on one thread and launched from
    another then sees the same dispatch on both.  An explicit ds_set_option still has to be made on the launching thread."""
    lib = lib or _lib
    _tls.env_applied = True
    for kv in filter(None, os.environ.get("DS_OPTIONS", "").split(",")):   # A/B runs: DS_OPTIONS=key=value,key=value
        k, _, val = kv.partition("=")
        if lib.ds_set_option(k.strip().encode(), int(val)) != 0:
            raise DiffSenseiHipError(lib.ds_last_error().decode())
    v = os.environ.get("DS_GEMM_VARIANT")
    if v is not None:  # tuning/A-B knob only; 0 = the library's own choice
        if lib.ds_set_option(b"gemm_variant", int(v)) != 0:
            raise DiffSenseiHipError(lib.ds_last_error().decode())


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ds_last_error().decode(errors="replace")
        raise DiffSenseiHipError(f"{what or 'diffsensei_hip'} failed (rc={rc}): {msg}")
