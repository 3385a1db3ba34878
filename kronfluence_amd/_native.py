"""ctypes binding of ``libkronfluence_hip.so`` (C ABI: ``include/kronfluence_hip.h``).

The product has no CPU fallback: loading fails loudly if the library has not been built, and every
compute entry point fails loudly when no MI355X is visible.  ``torch`` is imported first so that the
library binds to the HIP runtime PyTorch already loaded (same ``libamdhip64.so.7`` SONAME) and its
kernels run on PyTorch's streams.
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch  # noqa: F401  (must precede the dlopen below)

LIB_NAME = "libkronfluence_hip.so"
ABI_VERSION = 14

KF_F32, KF_BF16, KF_F16, KF_F64, KF_I64, KF_I32, KF_U8 = range(7)

_DTYPE_CODES = {
    torch.float32: KF_F32, torch.bfloat16: KF_BF16, torch.float16: KF_F16, torch.float64: KF_F64,
    torch.int64: KF_I64, torch.int32: KF_I32, torch.uint8: KF_U8, torch.bool: KF_U8,
}


class KfError(RuntimeError):
    """Raised for any non-zero ``kf_status``."""


class kf_view(ctypes.Structure):
    _fields_ = [
        ("p", ctypes.c_void_p), ("dtype", ctypes.c_int),
        ("batch_stride", ctypes.c_int64), ("row_stride", ctypes.c_int64), ("k_stride", ctypes.c_int64),
        ("rows", ctypes.c_int64), ("depth", ctypes.c_int64),
        ("ones_row", ctypes.c_int), ("ones_k", ctypes.c_int), ("square", ctypes.c_int),
        ("k_tile_stride", ctypes.c_int64),
    ]


_i, _i64, _p, _f, _d = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); mirrors include/kronfluence_hip.h declaration by declaration.
SIGNATURES = {
    "kf_abi_version": (_i, []),
    "kf_status_string": (ctypes.c_char_p, [_i]),
    "kf_device_count": (_i, []),
    "kf_syrk_accum": (_i, [_p, _i64, _p, _i, _i64, _i64, _i64, _i64, _i64, _i64, _p, _i, _i, _f, _p, _p]),
    "kf_syrk_rows_workspace_bytes": (_i64, [_i64, _i64, _i64, _i]),
    "kf_syrk_rows_bf16": (_i, [_p, _i64, _p, _i64, _i64, _i64, _p, _i, _i, _f, _p, _i64, _p]),
    "kf_syrk_rows_f32_workspace_bytes": (_i64, [_i64, _i64]),
    "kf_syrk_rows_f32": (_i, [_p, _i64, _p, _i64, _i64, _p, _i, _i, _f, _p, _i64, _p]),
    "kf_syrk_planes_workspace_bytes": (_i64, [_i64]),
    "kf_syrk_planes_bf16": (_i, [_p, _i64, _p, _i64, _i64, _i64, _f, _p, _i64, _p]),
    "kf_conv2d_cov_workspace_bytes": (_i64, [_i64] * 4 + [_i] * 8),
    "kf_conv2d_cov_accum": (_i, [_p, _i64, _p] + [_i64] * 4 + [_i] * 8 + [_f, _p, _i64, _p]),
    "kf_conv2d_cov_small": (_i, [_p, _i64, _p, _i, _i64, _i64, _i64, _i64] + [_i] * 9 + [_f, _p]),
    "kf_im2col": (_i, [_p, _i, _p, _i, _i64, _i64, _i64, _i64] + [_i] * 10 + [_p]),
    "kf_gemm": (_i, [_p, _i64, _i64, ctypes.POINTER(kf_view), ctypes.POINTER(kf_view), _i64, _f, _f, _p, _i64, _p]),
    "kf_gemm_out": (_i, [_p, _i, _i64, _i64, ctypes.POINTER(kf_view), ctypes.POINTER(kf_view), _i64, _f, _p]),
    "kf_gemm_bias_out": (_i, [_p, _i64, ctypes.POINTER(kf_view), ctypes.POINTER(kf_view), _p, _i64, _p]),
    "kf_eigh_workspace_bytes": (_i64, [_i64]),
    "kf_eigh_f64": (_i, [_p, _i, _d, _d, _i64, _p, _p, _p, _i64, _i, ctypes.POINTER(_i), _p]),
    "kf_eigh_stats": (None, [ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i]),
    "kf_lambda_accum": (_i, [_p, _i64, _p, _p, _i64, _i, _i64, _i64, _i64, _i64, _f, _p]),
    "kf_rotate_rows_transposed_bf16": (_i, [_p, _p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _p]),
    "kf_lambda_rows_accum": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _p]),
    "kf_lambda_conv2d_channels": (_i64, [_i64] * 5 + [_i] * 8),
    "kf_lambda_conv2d_workspace_bytes": (_i64, [_i64] * 5 + [_i] * 8),
    "kf_lambda_conv2d_accum": (_i, [_p, _i64, _p, _p] + [_i64] * 5 + [_i] * 8 + [_p, _i64, _i64, _f, _p, _i64, _p]),
    "kf_inv_lambda": (_i, [_p, _p, _i64, _d, _d, _p, _p]),
    "kf_precondition_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "kf_precondition": (_i, [_p, _i, _i64, _p, _p, _i, _i64, _i64, _i64, _i64, _i, _p, _p, _p, _f, _p, _p, _p, _i64, _p, _i64, _p]),
    "kf_precondition_bf16": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _p, _p, _p, _i64, _p, _p, _f, _p, _i64, _p]),
    "kf_pairwise_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "kf_pairwise_score": (_i, [_p, _i64, _p, _i, _i64, _i64, _p, _p, _i, _i64, _i64, _i64, _i64, _i, _f, _p, _i64, _p]),
    "kf_pairwise_conv2d_workspace_bytes": (_i64, [_i64] * 5 + [_i] * 8),
    "kf_pairwise_score_conv2d": (_i, [_p, _i64, _p, _i64, _p, _p] + [_i64] * 5 + [_i] * 8 + [_f, _p, _i64, _p]),
    "kf_pairwise_rows_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "kf_pairwise_score_rows": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _f, _p, _i64, _p]),
    "kf_pairwise_score_rows2": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _f, _p, _i64, _p]),
    "kf_eigh_small_batched": (_i, [_p, _i64, _i, _p, _p, _i, _f, _i, _p]),
    "kf_rowwise_dot": (_i, [_p, _p, _i, _p, _i, _p, _i64, _i64, _f, _i, _p]),
    "kf_lowrank_rows_dot": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _f, _p]),
    "kf_mul_bcast": (_i, [_p, _p, _i, _p, _i64, _i64, _f, _p]),
    "kf_cast": (_i, [_p, _i, _p, _i, _i64, _p]),
}

_lib: Optional[ctypes.CDLL] = None


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def lib() -> ctypes.CDLL:
    """Loads the HIP library once; raises if it is missing or its ABI differs."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise KfError(
                f"{path} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
                "g.build()\"` (or kronfluence_amd/csrc/build.sh). There is no CPU fallback."
            )
        handle = ctypes.CDLL(path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if an export is missing
            fn.restype, fn.argtypes = restype, argtypes
        if handle.kf_abi_version() != ABI_VERSION:
            raise KfError(f"{LIB_NAME} ABI {handle.kf_abi_version()} != binding ABI {ABI_VERSION}; rebuild.")
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise KfError(f"{what} failed: {lib().kf_status_string(status).decode()} ({status})")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODES[dtype]
    except KeyError as exc:
        raise KfError(f"dtype {dtype} is not supported by the HIP extension") from exc


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise KfError(
            f"`{name}` lives on {t.device}; the EK-FAC hot path runs only on an MI355X (there is no CPU fallback)."
        )


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
