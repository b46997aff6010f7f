"""ctypes binding of libframeino_hip.so (C ABI declared in include/frameino_hip.h).

The product path has no fallback: if the shared library is absent this module raises, and every
wrapper raises RuntimeError(fino_last_error()) on a non-zero return code."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# FINO_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("FINO_LIB_PATH") or os.path.join(_HERE, "lib", "libframeino_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "frameino_hip.h")

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "fino_version": [],
    "fino_last_error": [],
    "fino_tune_set": [c_int, c_int],
    "fino_tune_get": [c_int],
    "fino_adaln_modulate": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_void_p,
                            c_float, c_int, c_void_p],
    "fino_layernorm": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_void_p, c_void_p, c_float, c_int, c_void_p],
    "fino_gated_residual": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64,
                            c_void_p, c_int, c_void_p],
    "fino_gated_residual_staged": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64,
                                   c_void_p, c_int, c_void_p],
    "fino_layernorm_zero": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_i64, c_void_p, c_float, c_int, c_void_p],
    "fino_rmsnorm_rope": [c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int,
                          c_void_p],
    "fino_rmsnorm_rope_scaled": [c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p, c_void_p, c_int, c_float,
                                 c_int, c_void_p],
    "fino_qkv_rmsnorm_rope": [c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p, c_float, c_void_p, c_void_p, c_int,
                              c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "fino_rmsnorm_rope_scatter": [c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p, c_void_p, c_int, c_float,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "fino_diag_mfma_peak": [c_int, c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double), c_void_p],
    "fino_headnorm_rope": [c_void_p, c_int, c_i64, c_int, c_int, c_i64, c_i64, c_void_p, c_void_p, c_float,
                           c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "fino_headnorm_rope_scaled": [c_void_p, c_int, c_i64, c_int, c_int, c_i64, c_i64, c_void_p, c_void_p, c_float,
                                  c_void_p, c_void_p, c_i64, c_float, c_int, c_void_p],
    "fino_attn_fwd": [c_void_p] * 4 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 12 + [c_float, c_int, c_void_p],
    "fino_attn_fwd_ws": [c_void_p] * 4 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 12 +
                        [c_float, c_int, c_void_p, c_i64, c_void_p],
    "fino_attn_workspace_bytes": [c_int, c_int, c_i64, c_i64, c_int],
    "fino_attn_tail_supported": [c_int, c_int, c_i64, c_i64, c_int],
    "fino_attn_fwd_tail": [c_void_p] * 4 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 12 + [c_float, c_int, c_void_p, c_void_p,
                                                                                            c_void_p],
    "fino_attn_probs_supported": [c_int, c_int, c_i64, c_i64, c_int],
    "fino_attn_probs": [c_void_p] * 3 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 6 + [c_int, c_i64, c_i64, c_float, c_int,
                                                                                         c_void_p, c_void_p, c_void_p, c_i64,
                                                                                         c_void_p, c_void_p],
    "fino_row_rrms": [c_void_p, c_i64, c_int, c_i64, c_float, c_void_p, c_int, c_void_p],
    "fino_attn_fp8_kv_bytes": [c_int, c_int, c_i64, c_int],
    "fino_attn_fwd_fp8": [c_void_p] * 4 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 8 + [c_float, c_int, c_int, c_void_p,
                                                                                           c_i64, c_void_p],
    "fino_attn_partial_bytes": [c_int, c_int, c_i64, c_int],
    "fino_attn_partial": [c_void_p] * 3 + [c_int, c_int, c_i64, c_i64, c_int] + [c_i64] * 9 + [c_float, c_int, c_void_p,
                                                                                               c_i64, c_void_p],
    "fino_attn_merge": [c_void_p, c_int, c_int, c_i64, c_int, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_int,
                        c_void_p],
    "fino_mxfp8_scale_bytes": [c_i64, c_i64],
    "fino_quantize_mxfp8": [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p],
    "fino_gemm_mxfp8": [c_void_p] * 6 + [c_i64] * 4 + [c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "fino_gemm_mxfp8_q": [c_void_p] * 7 + [c_i64] * 3 + [c_int, c_int, c_void_p],
    "fino_traj_paint": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "fino_traj_blur_quantize": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "fino_groupnorm_workspace_bytes": [c_int],
    "fino_groupnorm_cl": [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p, c_float, c_void_p, c_void_p] +
                         [c_int] * 4 + [c_void_p, c_i64, c_int, c_void_p],
    "fino_avg_pool_time2": [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "fino_resize_area_pad_u8": [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p],
    "fino_u8_hwc_to_chw_unit": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "fino_gemm": [c_void_p] * 4 + [c_i64] * 6 + [c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "fino_gemm_split_n": [c_void_p] * 4 + [c_i64] * 6 + [c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p,
                          c_i64, c_i64, c_int, c_void_p],
    "fino_ln_mxfp8": [c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                      c_void_p, c_float, c_int, c_void_p],
    "fino_gemm_blocked_a": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_i64, c_i64,
                            c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p],
    "fino_gemm_plan": [c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_int)],
    "fino_skinny_linear": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_int, c_int, c_void_p],
    "fino_patchify": [c_void_p, c_void_p] + [c_int] * 7 + [c_i64, c_int, c_void_p],
    "fino_unpatchify": [c_void_p, c_void_p] + [c_int] * 7 + [c_i64, c_int, c_void_p],
    "fino_wan_model_input": [c_void_p] * 5 + [c_int] * 6 + [c_void_p],
    "fino_cfg_euler_step": [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_float, c_void_p, c_int, c_int, c_void_p],
    "fino_cfg_unipc_step": [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_int, c_void_p],
    "fino_cfg_vpred_step": [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_int, c_int, c_void_p],
    "fino_cfg_dpm_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_int, c_int, c_void_p],
    "fino_conv3d": [c_void_p] * 4 + [c_int] * 19 + [c_void_p, c_void_p, c_int, c_void_p],
    "fino_rmsnorm_silu_cl": [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "fino_split_bf16": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_int, ctypes.c_uint, c_void_p],
    "fino_rmsnorm_silu_cl_f32": [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_int, ctypes.c_uint, c_void_p],
    "fino_conv3d_split": [c_void_p] * 4 + [c_int] * 20 + [c_void_p, c_void_p, c_void_p],
    "fino_vae_blend_tiles": [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p],
    "fino_softmax_rows": [c_void_p, c_i64, c_int, c_i64, c_float, c_int, c_void_p],
    "fino_dup_up3d_add": [c_void_p] * 3 + [c_int] * 10 + [c_void_p],
    "fino_avg_down3d_add": [c_void_p] * 3 + [c_int] * 10 + [c_void_p],
    "fino_vae_unpatchify_clamp": [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    "fino_vae_patchify": [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
}
_RESTYPES = {"fino_last_error": ctypes.c_char_p, "fino_attn_workspace_bytes": c_i64, "fino_mxfp8_scale_bytes": c_i64,
             "fino_groupnorm_workspace_bytes": c_i64,
             "fino_attn_partial_bytes": c_i64, "fino_attn_fp8_kv_bytes": c_i64}


# the FINO_VERSION this table (argument lists, tune-knob meanings) was written for: a stale library found through
# FINO_LIB_PATH would otherwise fail late (AttributeError on a new symbol) or silently misread an argument
ABI_VERSION = 103


def declared_symbols(header_path=HEADER_PATH):
    """Every function the public header declares (used by the CPU test that checks the .so exports them)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fino_[a-z0-9_]+)\s*\(", text)))


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError(
            f"frameino_amd: HIP library not found at {path}. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C frameino_amd/csrc`). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(path)
    lib.fino_version.restype = c_int
    if lib.fino_version() < 0 and os.environ.get("FINO_ALLOW_EXPERIMENT") != "1":
        raise RuntimeError(
            f"frameino_amd: {path} is a timing-EXPERIMENT build (compiled with -DFINO_EXPERIMENT: its kernels may skip "
            f"work and return wrong results; fino_version() = {lib.fino_version()}).  It is refused as the product library; "
            f"set FINO_ALLOW_EXPERIMENT=1 for the tools/ scripts that time it.")
    if abs(lib.fino_version()) != ABI_VERSION:
        raise RuntimeError(
            f"frameino_amd: {path} reports fino_version() = {lib.fino_version()}, this package binds ABI version "
            f"{ABI_VERSION} (include/frameino_hip.h: FINO_VERSION).  Rebuild it (`make -C frameino_amd/csrc`).")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().fino_last_error().decode()}")
