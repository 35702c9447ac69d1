"""ctypes binding of the C-ABI in include/ldm_amd.h (libldm_amd.so).

The library is the product: there is no Python or CPU fallback.  If the shared object is missing or an
entry point fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE the library is dlopen'ed: the PyTorch-ROCm wheel ships its own HIP runtime
# (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and the kernels must run in that runtime, because the device
# pointers and streams they are given are torch's.  Loaded first, torch's copy satisfies this library's
# DT_NEEDED by SONAME; loaded second, a second runtime (/opt/rocm) would come up beside it and every launch
# would fail with "no ROCm-capable device is detected".
import torch  # noqa: F401

F32, BF16 = 0, 1

EPI_BIAS, EPI_ROWVEC, EPI_RESIDUAL, EPI_SILU, EPI_GEGLU, EPI_SWIGLU, EPI_ROWSTATS, EPI_LNFOLD = 1, 2, 4, 8, 16, 32, 64, 128
ABI_VERSION = 3

# LDM_AMD_LIB overrides the path (A/B runs of two builds of the library; there is still no fallback).
_LIB_PATH = os.environ.get("LDM_AMD_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "lib", "libldm_amd.so")

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float
PF = C.POINTER(C.c_float)
PI64 = C.POINTER(C.c_int64)


class LdmError(RuntimeError):
    pass


class ConvGemmDesc(C.Structure):
    """struct LdmConvGemm (include/ldm_amd.h)."""
    _fields_ = [
        ("a0", c_void_p), ("a1", c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("batch", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32),
        ("w", c_void_p), ("n", C.c_int32),
        ("epilogue", C.c_int32),
        ("bias", c_void_p), ("rowvec", c_void_p), ("rowvec_stride", C.c_int32),
        ("residual", c_void_p), ("out", c_void_p),
        ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("workspace", c_void_p), ("workspace_bytes", C.c_int64),
        ("rowstats", c_void_p), ("ln_stats", c_void_p), ("ln_colsum", c_void_p),
        ("ln_segments", C.c_int32), ("ln_eps", C.c_float),
        ("glu_preact", c_void_p),
    ]


class PackJob(C.Structure):
    """struct LdmPackJob (include/ldm_amd.h)."""
    _fields_ = [("src", c_void_p), ("fwd", c_void_p), ("dgrad", c_void_p), ("dgrad_ld", C.c_int64),
                ("cout", C.c_int32), ("cin", C.c_int32), ("glu_pairs", C.c_int32),
                ("blocks_x", C.c_int32), ("blocks_y", C.c_int32), ("first_block", C.c_int32)]


# name -> argtypes; every function returns int except the two noted below
_SIGNATURES = {
    "ldm_ddpm_build_tables": [c_int, c_int, PF, PF, PF, PF, PF, PF, PF],
    "ldm_ddpm_set_timesteps": [c_int, c_int, c_int, c_int, PI64],
    "ldm_ddpm_step_coefficients": [PF, PF, PF, c_int, c_int, c_int, PF],
    "ldm_ddpm_step": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, PF, c_void_p],
    "ldm_ddpm_step_cfg": [c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                          c_i64, PF, c_void_p],
    "ldm_ddpm_add_noise": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_i64, c_void_p],
    "ldm_pack_weight": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ldm_conv_gemm": [C.POINTER(ConvGemmDesc), c_void_p],
    "ldm_conv_gemm_tile": [C.POINTER(ConvGemmDesc), C.POINTER(C.c_int)],
    "ldm_conv_wgrad": [C.POINTER(ConvGemmDesc), c_void_p, c_void_p, c_int, c_void_p, C.c_longlong, c_void_p],
    "ldm_sum2x2": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ldm_colsum": [c_void_p, c_int, C.c_longlong, C.c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "ldm_glu_forward": [c_void_p, c_void_p, c_int, C.c_longlong, c_int, c_int, c_void_p],
    "ldm_glu_backward": [c_void_p, c_void_p, c_void_p, c_int, C.c_longlong, c_int, c_int, c_void_p],
    "ldm_opt_chunk_elements": [],
    "ldm_grad_norm_clip": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, C.c_float, c_void_p, c_void_p, c_void_p],
    "ldm_adamw_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, C.c_double, C.c_double,
                       C.c_double, C.c_double, C.c_double, C.c_longlong, c_void_p, c_void_p],
    "ldm_mse_loss": [c_void_p, c_int, c_void_p, C.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p],
    "ldm_train_input_prep": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_i64,
                             C.c_float, C.c_float, C.c_float, c_void_p, c_void_p, c_void_p],
    "ldm_cond_dropout": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_i64, c_void_p],
    "ldm_gaussian_noise": [C.c_uint64, c_i64, c_i64, c_i64, c_i64, c_void_p, c_void_p],
    "ldm_pack_weight_pair": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, C.c_longlong, c_void_p],
    "ldm_pack_weight_pair_batch": [c_void_p, c_int, c_int, c_int, c_void_p],
    "ldm_add": [c_void_p, C.c_longlong, c_void_p, C.c_longlong, c_void_p, c_int, C.c_longlong, c_int, c_void_p],
    "ldm_silu_backward": [c_void_p, c_void_p, c_void_p, c_int, C.c_longlong, c_void_p],
    "ldm_groupnorm_backward": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "ldm_layernorm_backward": [c_void_p, c_void_p, c_int, C.c_longlong, c_int, c_float, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "ldm_groupnorm_workspace_floats": [c_int, c_int],
    "ldm_groupnorm_stats": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                            c_void_p, c_void_p],
    "ldm_groupnorm_apply": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                            c_void_p, c_int, c_void_p, c_void_p],
    "ldm_groupnorm": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                      c_void_p, c_void_p, c_void_p, c_void_p],
    "ldm_groupnorm_noalloc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p, c_void_p],
    "ldm_layernorm": [c_void_p, c_int, c_i64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "ldm_attention": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                      c_int, c_int, c_void_p],
    "ldm_attention_train": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_void_p, c_void_p],
    "ldm_attention_backward": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                               c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_void_p],
    "ldm_timestep_embedding": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "ldm_silu": [c_void_p, c_void_p, c_int, c_i64, c_void_p],
    "ldm_conv_in": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                    c_void_p],
    "ldm_conv_out": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                     c_void_p],
    "ldm_softmax_rows": [c_void_p, C.c_longlong, C.c_longlong, c_int, c_int, c_float, c_void_p, c_int, C.c_longlong, c_void_p],
    "ldm_conv1x1_nchw": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, C.c_longlong, c_void_p],
    "ldm_image_to_uint8": [c_void_p, c_int, c_void_p, C.c_longlong, c_void_p],
}

# host-only size queries returning long long
_LONGLONG_QUERIES = {
    "ldm_colsum_workspace_floats": [C.c_longlong, c_int],
    "ldm_groupnorm_backward_workspace_floats": [c_int, c_int, c_int],
    "ldm_layernorm_backward_workspace_floats": [c_int],
    "ldm_mse_workspace_floats": [C.c_longlong],
}

EXPORTED_SYMBOLS = (["ldm_abi_version", "ldm_last_error", "ldm_conv_wgrad_workspace_bytes"] + list(_SIGNATURES)
                    + list(_LONGLONG_QUERIES))

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libldm_amd.so (once).  Raises LdmError if it is not built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LdmError(f"{_LIB_PATH} not found: build it with `make -C latent-diffusion_amd/csrc` "
                       "(or __graft_entry__.build()); this package has no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    lib.ldm_abi_version.restype = c_int
    lib.ldm_abi_version.argtypes = []
    lib.ldm_last_error.restype = C.c_char_p
    lib.ldm_last_error.argtypes = []
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = argtypes
    lib.ldm_conv_wgrad_workspace_bytes.restype = C.c_longlong
    lib.ldm_conv_wgrad_workspace_bytes.argtypes = [C.POINTER(ConvGemmDesc)]
    for name, argtypes in _LONGLONG_QUERIES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_longlong
        fn.argtypes = argtypes
    if lib.ldm_abi_version() != ABI_VERSION:
        raise LdmError(f"ABI version mismatch: library reports {lib.ldm_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    """Turn a negative status into an exception carrying the library's message."""
    if rc != 0:
        msg = load().ldm_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg or f"{what}: invalid argument")
        raise LdmError(f"{what}: error {rc}: {msg}")
