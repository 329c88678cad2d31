"""ctypes binding of libicl_hip.so (C ABI: include/icl_hip.h).

The product path has no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  `_use_library_for_tests` exists only so that tests/ can point the same binding at the
CPU-fiber emulation build of the very same kernels (tests/hipemu) in the GPU-less build container.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  — must be loaded BEFORE libicl_hip.so: both link libamdhip64, and the process must end up with
#                       torch's copy of the HIP runtime only (two runtimes: "no ROCm-capable device is detected" at launch)

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libicl_hip.so")

_lib = None
_lib_path = None
_host_pointers_ok = False  # True only for the test-only emulation library

P, I, L, F = c_void_p, c_int, c_int64, c_float

_SIGNATURES = {
    "icl_abi_version": (c_int, []),
    "icl_last_error": (c_char_p, []),
    "icl_last_kernel_name": (c_char_p, []),
    "icl_conv3d_packed_elems": (c_int64, [I, I, I, I]),
    "icl_conv3d_pack_weights": (c_int, [P, P, I, I, I, I, P]),
    "icl_conv3d_pack_weights_both": (c_int, [P, P, P, I, I, I, P]),
    "icl_conv3d_pack_weights_multi": (c_int, [P, P, P, P, P, P, I, P]),
    "icl_conv3d_fwd_ws_bytes": (c_int64, [I, I, I, I, I, I, I]),
    "icl_conv3d_fwd": (c_int, [P, P, P, P, P, I, I, I, I, I, I, I, L, L, P]),
    "icl_scalar_combine": (c_int, [P, P, I, I, P, P]),
    "icl_conv3d_split_ws_bytes": (c_int64, [I, I]),
    "icl_conv3d_split_weights_multi": (c_int, [P, P, P, P, I, P]),
    "icl_conv3d_fwd_presplit": (c_int, [P, P, P, P, I, I, I, I, I, I, L, L, P]),
    "icl_conv3d_fwd_presplit_ws_bytes": (c_int64, [I, I, I, I, I, I]),
    "icl_conv3d_fwd_presplit_ws": (c_int, [P, P, P, P, P, I, I, I, I, I, I, L, L, P]),
    "icl_conv3d_fwd_stats_slots": (c_int, [I, I, I, I, I, I]),
    "icl_conv3d_fwd_presplit_stats": (c_int, [P, P, P, P, P, I, I, I, I, I, I, L, L, P]),
    "icl_norm_fwd_given_stats": (c_int, [P, P, P, P, P, P, P, P, P, I, I, L, I, I, F, F, P, I, P]),
    "icl_conv3d_wgrad_ws_bytes": (c_int64, [I, I, I, I]),
    "icl_conv3d_wgrad": (c_int, [P, P, P, P, P, I, I, I, I, I, I, I, L, L, P]),
    "icl_conv3d_wgrad_slabs": (c_int, [P, P, P, I, I, I, I, I, I, I, L, L, P, P]),
    "icl_conv3d_wgrad_reduce_multi": (c_int, [P, P, P, P, P, P, I, P]),
    "icl_norm_ws_bytes": (c_int64, [I, I, L]),
    "icl_norm_fwd": (c_int, [P, P, P, P, P, P, P, P, I, I, L, I, I, I, F, F, P, P]),
    "icl_norm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, L, I, I, I, P, P]),
    "icl_norm_res_fwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, L, I, I, I, F, F, P, P]),
    "icl_norm_res_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I, I, L, I, I, I, P, P]),
    "icl_rstd_from_var": (c_int, [P, P, I, F, P]),
    "icl_norm_finalize_stats": (c_int, [P, I, I, I, F, P, P, P, P]),
    "icl_norm_apply": (c_int, [P, P, P, I, I, L, P]),
    "icl_maxpool2_fwd_norm": (c_int, [P, P, P, P, L, I, I, I, I, P]),
    "icl_upsample2x_concat_norm": (c_int, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "icl_conv1x1_dropout_norm": (c_int, [P, P, P, P, P, I, I, I, L, I, I, ctypes.c_uint32, F, P, P]),
    "icl_conv1x1_wgrad_dropout_norm": (c_int, [P, P, P, P, P, P, I, I, I, L, L, ctypes.c_uint32, F, P, P]),
    "icl_maxpool2_fwd": (c_int, [P, P, P, L, I, I, I, I, P]),
    "icl_maxpool2_bwd": (c_int, [P, P, P, L, I, I, I, I, P]),
    "icl_maxpool2_bwd_add": (c_int, [P, P, P, P, I, I, I, I, I, I, L, P]),
    "icl_trilinear_fwd": (c_int, [P, P, I, I, I, I, I, I, I, I, L, I, P]),
    "icl_trilinear_bwd_ws_bytes": (c_int64, [I, I, I, I, I, I, I, I]),
    "icl_trilinear_bwd": (c_int, [P, P, P, I, I, I, I, I, I, I, I, L, I, P]),
    "icl_copy_rows": (c_int, [P, P, L, L, L, L, P]),
    "icl_concat2": (c_int, [P, L, P, L, P, P]),
    "icl_dwconv3_fwd": (c_int, [P, P, P, I, I, I, I, I, I, P]),
    "icl_dwconv3_wgrad_ws_bytes": (c_int64, [I, I, I, I, I]),
    "icl_dwconv3_wgrad": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "icl_conv1x1_small": (c_int, [P, P, P, P, I, I, I, L, I, I, P]),
    "icl_conv1x1_dropout": (c_int, [P, P, P, P, I, I, I, L, I, I, I, ctypes.c_uint32, F, P, P]),
    "icl_conv1x1_wgrad_dropout": (c_int, [P, P, P, P, P, I, I, I, L, L, ctypes.c_uint32, F, P, P]),
    "icl_dropout": (c_int, [P, P, L, ctypes.c_uint32, F, P, P]),
    "icl_drop_path": (c_int, [P, P, L, L, ctypes.c_uint32, F, P, P]),
    "icl_drop_path_add": (c_int, [P, P, P, L, L, ctypes.c_uint32, F, P, P]),
    "icl_loss_fwd": (c_int, [P, P, P, P, P, P, I, I, L, I, I, P]),
    "icl_window_attn_bias_elems": (c_int64, [I, I]),
    "icl_depth_to_space2": (c_int, [P, P, I, I, I, I, I, L, P]),
    "icl_space_to_depth2": (c_int, [P, P, I, I, I, I, I, L, P]),
    "icl_colsum_multi": (c_int, [P, P, P, P, I, P]),
    "icl_gather_rows": (c_int, [P, P, P, L, L, L, I, P]),
    "icl_gather_rows_sum2": (c_int, [P, P, P, L, L, L, I, P]),
    "icl_im2col3": (c_int, [P, P, I, I, I, I, I, P]),
    "icl_im2col3_planes": (c_int, [P, P, I, I, I, I, I, P]),
    "icl_conv3d_cin1_fwd": (c_int, [P, P, P, P, I, I, I, I, I, L, L, P]),
    "icl_conv3d_cin1_wgrad_ws_bytes": (c_int64, [I, I, I]),
    "icl_conv3d_cin1_wgrad": (c_int, [P, P, P, P, I, I, I, I, I, L, L, P]),
    "icl_col2im3": (c_int, [P, P, I, I, I, I, I, P]),
    "icl_linear_ws_bytes": (c_int64, [L, I, I, I]),
    "icl_linear_fwd": (c_int, [P, P, P, P, P, L, I, I, I, P]),
    "icl_linear_dgrad": (c_int, [P, P, P, P, L, I, I, P]),
    "icl_linear_bwd_small": (c_int, [P, P, P, P, P, P, L, I, I, P]),
    "icl_linear_wgrad_small": (c_int, [P, P, P, P, L, I, I, P]),
    "icl_linear_dgrad_sgd": (c_int, [P, P, P, P, P, P, L, I, I, F, F, F, I, P, P]),
    "icl_gemm_ws_bytes": (c_int64, [L, I, I, I]),
    "icl_gemm": (c_int, [P, P, P, P, P, L, I, I, L, L, L, I, I, I, I, L, L, L, P]),
    "icl_linear_wgrad_ws_bytes": (c_int64, [L, I, I]),
    "icl_linear_wgrad": (c_int, [P, P, P, P, P, L, I, I, P]),
    "icl_conv1x1_wgrad_ws_bytes": (c_int64, [I, L, I, I]),
    "icl_conv1x1_wgrad": (c_int, [P, P, P, P, P, I, I, I, L, L, L, P]),
    "icl_relpos_bias_fwd": (c_int, [P, P, P, I, I, I, P]),
    "icl_window_attn_fwd": (c_int, [P, P, P, P, P, I, I, I, I, I, F, P]),
    "icl_window_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "icl_window_attn_bwd_chunks": (c_int, [I, I, I, I, I]),
    "icl_relpos_bias_bwd_sum": (c_int, [P, I, P, P, P, L, I, I, P]),
    "icl_layernorm_fwd": (c_int, [P, P, P, P, P, P, L, I, F, P]),
    "icl_layernorm_bwd_ws_bytes": (c_int64, [L, I]),
    "icl_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, L, I, P]),
    "icl_gelu_fwd": (c_int, [P, P, L, P]),
    "icl_gelu_bwd": (c_int, [P, P, P, L, P]),
    "icl_attn_fwd": (c_int, [P, P, P, P, P, I, I, I, I, I, F, P]),
    "icl_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "icl_attn_bwd_ws_bytes": (c_int64, [I, I, I, I, I]),
    "icl_attn_bwd_ws": (c_int, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "icl_crop_rotflip": (c_int, [P, P, P, I, P, P, I, I, I, P]),
    "icl_sgd_step": (c_int, [P, P, P, L, F, F, F, I, P, P]),
    "icl_sgd_step_factored": (c_int, [P, P, P, P, I, I, I, F, F, F, I, P, P]),
    "icl_sgd_step_factored_narrow": (c_int, [P, P, P, P, I, I, I, F, F, F, I, P, I, P]),
    "icl_sgd_factored_split_ws_bytes": (c_int64, [I, I, I]),
    "icl_sgd_step_factored_split": (c_int, [P, P, P, P, P, I, I, I, F, F, F, I, P, P]),
    "icl_sgd_step_multi": (c_int, [P, P, P, P, I, F, F, F, I, P, P]),
    "icl_loss_bwd": (c_int, [P, P, P, P, P, P, P, P, I, I, L, I, I, P]),
    "icl_loss_fwd_multi": (c_int, [P, I, P]),
    "icl_loss_bwd_multi": (c_int, [P, I, P]),
    "icl_qchain_stage": (c_int, [P, P]),
    "icl_qchain_wgrad": (c_int, [P, P, P, P, P, P, P, P, I, P]),
}

EXPORTS = tuple(_SIGNATURES)


def _load(path: str):
    if not os.path.exists(path):
        raise RuntimeError(
            f"icl_amd: HIP library {path} not found — build it with `python -m icl_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    global _lib, _lib_path
    if _lib is None:
        _lib_path = os.environ.get("ICL_HIP_LIB", DEFAULT_LIB)
        _lib = _load(_lib_path)
    return _lib


def lib_path():
    lib()
    return _lib_path


def host_pointers_ok() -> bool:
    return _host_pointers_ok


def _use_library_for_tests(path: str | None, host_pointers: bool = True):
    """TESTS ONLY: route the binding to another build of the same ABI (the CPU emulation)."""
    global _lib, _lib_path, _host_pointers_ok
    if path is None:
        _lib, _lib_path, _host_pointers_ok = None, None, False
        return
    _lib, _lib_path, _host_pointers_ok = _load(path), path, host_pointers


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().icl_last_error()
        raise RuntimeError(f"icl_hip {what} failed ({rc}): {msg.decode() if msg else ''}")
