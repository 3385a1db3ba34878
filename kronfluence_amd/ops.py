"""Tensor-level entry points of the hot path: thin, allocation-light wrappers that hand raw device
pointers and the current HIP stream to the C ABI (``include/kronfluence_hip.h``).

Nothing here computes on the host; every function raises ``KfError`` if its tensors are not on an
MI355X.  Accumulators are fp32 device tensors owned by the caller (the trackers); workspaces come
from PyTorch's caching allocator.
"""

from __future__ import annotations

import ctypes
import logging
import math
import os
from typing import Optional, Tuple

import torch
from torch import nn

from kronfluence_amd import _native as nat
from kronfluence_amd._native import kf_view


def _require(condition, what: str) -> None:
    """Argument validation that survives ``python -O`` (an ``assert`` does not): a violated precondition of the C ABI is a
    ``KfError`` like every other failure of this layer."""
    if not condition:
        raise nat.KfError(f"invalid argument: {what}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _contig(t: torch.Tensor, align: bool = True) -> torch.Tensor:
    """Input operand as the kernels want it: contiguous and 16-byte aligned (a contiguous VIEW into a larger tensor, e.g. a
    slice along the batch axis of an odd-sized sample, may start anywhere; the vector loads / LDS-DMA requests may not).
    ``align=False``: the consumer reads element-wise (the fp32 covariance / GEMM loaders), so an unaligned contiguous view is
    taken as it is instead of being cloned."""
    if t.is_contiguous() and (not align or t.data_ptr() % 16 == 0):
        return t
    return t.clone(memory_format=torch.contiguous_format)


def _vector_engines(t: torch.Tensor) -> bool:
    """Whether ``t``'s dtype is consumed by the 16-byte vector / LDS-DMA engines (bf16, fp16)."""
    return t.element_size() == 2


def view(t: torch.Tensor, batch_stride: int, row_stride: int, k_stride: int, rows: int, depth: int,
         ones_row: bool = False, ones_k: bool = False, square: bool = False, k_tile_stride: int = 0) -> kf_view:
    """Strided operand view over the STORAGE of ``t`` (strides in elements); ``t`` must be contiguous."""
    _require(t.is_contiguous(), "kf_view describes raw storage; pass a contiguous tensor")
    return kf_view(t.data_ptr(), nat.dtype_code(t.dtype), batch_stride, row_stride, k_stride, rows, depth,
                   int(ones_row), int(ones_k), int(square), k_tile_stride)


def k_tile_major(p: torch.Tensor) -> torch.Tensor:
    """``[rows, D] -> [D/64, rows, 64]`` (contiguous): the layout the bf16 score contraction streams best
    (see ``kf_view.k_tile_stride``).  ``D`` must be a multiple of 64."""
    rows = p.shape[0]
    flat = p.reshape(rows, -1)
    _require(flat.shape[1] % 64 == 0, 'flat.shape[1] % 64 == 0')
    return flat.view(rows, flat.shape[1] // 64, 64).transpose(0, 1).contiguous()


# ---------------------------------------------------------------------------------------------
# Stage 1: covariance
# ---------------------------------------------------------------------------------------------
def syrk_accum(cov: torch.Tensor, x: torch.Tensor, n_rows: int, d_in: int, rows_inner: int, outer_stride: int,
               row_stride: int, col_stride: int, mask: Optional[torch.Tensor] = None, append_ones: bool = False,
               alpha: float = 1.0, count: Optional[torch.Tensor] = None) -> None:
    """``cov += alpha * X'^T X'`` (kf_syrk_accum).  ``cov``: fp32 ``[d, d]`` device tensor."""
    nat.require_device(cov, "cov")
    nat.require_device(x, "x")
    _require(cov.dtype == torch.float32 and cov.is_contiguous(), 'cov.dtype == torch.float32 and cov.is_contiguous()')
    if mask is not None:
        nat.require_device(mask, "mask")
        mask = _contig(mask)
        _require(mask.numel() == n_rows, 'mask.numel() == n_rows')
    if count is not None:
        nat.require_device(count, "count")
        _require(count.dtype == torch.int64, 'count.dtype == torch.int64')
    d = d_in + int(append_ones)
    # fp32 rows run on the exact-fp32 MFMA engine (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak): timed under their own name so that
    # bench.py prices them against THAT peak, not the bf16 one
    with _Timed("syrk_accum" if x.element_size() == 2 else "syrk_accum_f32", x.device, float(n_rows) * d * (d + 1),
                float(n_rows) * d_in * x.element_size()):
        nat.check(
            nat.lib().kf_syrk_accum(cov.data_ptr(), cov.shape[1], x.data_ptr(), nat.dtype_code(x.dtype), n_rows, d_in,
                                    rows_inner, outer_stride, row_stride, col_stride, _ptr(mask),
                                    nat.dtype_code(mask.dtype) if mask is not None else 0, int(append_ones), alpha,
                                    _ptr(count), nat.stream_ptr(x.device)),
            "kf_syrk_accum",
        )


# The staged covariance kernels keep a [d_pad, d_pad] fp32 matrix in their workspace and finalize it with one workgroup row
# per matrix row (gridDim.y <= 65535): wider layers (a vocabulary-sized head) use kf_syrk_accum, which writes C directly.
COV_STAGED_MAX_DIM = 32768


def _syrk_rows_bf16(cov: torch.Tensor, x: torch.Tensor, mask: Optional[torch.Tensor], has_bias: bool, alpha: float) -> bool:
    """bf16 ``[b, T, d]`` rows of a sequence layer on the LDS-DMA covariance kernel (exact bf16 products, fp32 accumulation);
    ``False`` when the shape / dtype is not eligible.  Integer / bool masks: every row (and its bias one) is multiplied by its
    mask value inside the transposition kernel (0 and 1 exactly; other weights rounded to bf16 like the reference's ``mul_``)."""
    d_in = x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and x.shape[1] % 64 == 0 and d_in % 8 == 0 and d_in >= 64
            and d_in < COV_STAGED_MAX_DIM and 0 < x.shape[0] <= 65535 and x.data_ptr() % 16 == 0
            and (mask is None or mask.dtype in (torch.int64, torch.int32, torch.uint8, torch.bool))):
        return False
    b, t = x.shape[0], x.shape[1]
    mask = _contig(mask) if mask is not None else None
    ws_bytes = nat.lib().kf_syrk_rows_workspace_bytes(b, t, d_in, int(has_bias))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    d = d_in + int(has_bias)
    with _Timed("syrk_accum", x.device, float(b * t) * d * (d + 1), float(b * t) * d_in * 2):
        nat.check(
            nat.lib().kf_syrk_rows_bf16(cov.data_ptr(), cov.shape[1], x.data_ptr(), b, t, d_in, _ptr(mask),
                                        nat.dtype_code(mask.dtype) if mask is not None else 0, int(has_bias), alpha,
                                        ws.data_ptr(), ws_bytes, nat.stream_ptr(x.device)),
            "kf_syrk_rows_bf16",
        )
    return True


# fp32 rows of at least this many rows go through the exact three-term bf16 split (kf_syrk_rows_f32); KF_COV_F32_SPLIT=0: A/B, fallback
COV_F32_SPLIT_MIN_ROWS = 1024
COV_F32_SPLIT_MAX_ROWS = 65535 * 64


def _syrk_rows_f32(cov: torch.Tensor, x: torch.Tensor, mask: Optional[torch.Tensor], has_bias: bool, alpha: float) -> bool:
    """fp32 ``[..., d]`` rows (LayerNorm outputs under autocast with fp32 factors) on the bf16 MFMA engine through an EXACT split into
    three bf16 terms (kf_syrk_rows_f32: six bf16 products, fp32 accumulation; the dropped products are below one fp32 rounding);
    ``False`` when the shape / dtype is not eligible -- the exact-fp32 MFMA engine of ``syrk_accum`` takes those."""
    d_in = x.shape[-1]
    n = x.numel() // max(d_in, 1)
    # n: the split kernel's grid has one y-block per 64 rows (65 535 blocks at most); float masks (fractional weights: the count of
    # ``kf_syrk_accum`` is a float sum, this path counts integers) stay on the exact-fp32 engine
    if not (x.is_cuda and x.dtype == torch.float32 and d_in % 8 == 0 and 256 <= d_in < COV_STAGED_MAX_DIM
            and COV_F32_SPLIT_MIN_ROWS <= n <= COV_F32_SPLIT_MAX_ROWS
            and x.data_ptr() % 16 == 0 and os.environ.get("KF_COV_F32_SPLIT", "1") != "0"
            and (mask is None or mask.dtype in (torch.int64, torch.int32, torch.uint8, torch.bool))):
        return False
    mask = _contig(mask) if mask is not None else None
    ws_bytes = nat.lib().kf_syrk_rows_f32_workspace_bytes(n, d_in)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    d = d_in + int(has_bias)
    with _Timed("syrk_accum", x.device, float(n) * d * (d + 1), float(n) * d_in * 4):
        nat.check(
            nat.lib().kf_syrk_rows_f32(cov.data_ptr(), cov.shape[1], x.data_ptr(), n, d_in, _ptr(mask),
                                       nat.dtype_code(mask.dtype) if mask is not None else 0, int(has_bias), alpha, ws.data_ptr(),
                                       ws_bytes, nat.stream_ptr(x.device)),
            "kf_syrk_rows_f32",
        )
    return True


def linear_activation_cov(cov: torch.Tensor, count: torch.Tensor, x: torch.Tensor, mask: Optional[torch.Tensor],
                          has_bias: bool) -> None:
    """Flatten + mask + ones column + ``addmm_`` of module/linear.py:30-46 and tracker/factor.py:58, fused."""
    x = _contig(x, align=_vector_engines(x))
    d_in = x.shape[-1]
    n = x.numel() // d_in
    if mask is not None and mask.numel() != n:
        mask = None  # linear.py:33 -- the mask applies only when it matches the row count
    if _syrk_rows_bf16(cov, x, mask, has_bias, 1.0) or _syrk_rows_f32(cov, x, mask, has_bias, 1.0):
        count.add_(mask.sum().to(torch.int64) if mask is not None else n)
        return
    if mask is not None and mask.dtype not in (torch.float32, torch.int64, torch.uint8, torch.bool):
        # the reference multiplies by the mask whatever its dtype (``mul_``): bf16 / fp16 / int32 masks are widened
        mask = mask.to(torch.float32)
    syrk_accum(cov, x, n, d_in, max(n, 1), 0, d_in, 1, mask, has_bias, 1.0, count)


def linear_gradient_cov(cov: torch.Tensor, count: torch.Tensor, g: torch.Tensor, mask: Optional[torch.Tensor],
                        alpha: float = 1.0) -> None:
    """module/linear.py:48-54 + tracker/factor.py:93: gradient rows are never masked; the count is."""
    g = _contig(g, align=_vector_engines(g))
    d = g.shape[-1]
    n = g.numel() // d
    if not (_syrk_rows_bf16(cov, g, None, False, alpha) or _syrk_rows_f32(cov, g, None, False, alpha)):
        syrk_accum(cov, g, n, d, max(n, 1), 0, d, 1, None, False, alpha, None)
    if mask is not None and mask.numel() == n:
        count.add_(mask.sum().to(torch.int64))
    else:
        count.add_(n)


def conv_geometry(conv: nn.Conv2d) -> Tuple[int, int, int, int, int, int, int, int]:
    """Resolves string paddings like module/conv2d.py:46-53 (unequal padding is unsupported there too)."""
    from kronfluence_amd.utils.exceptions import UnsupportableModuleError

    k1, k2 = conv.kernel_size
    s1, s2 = conv.stride
    d1, d2 = conv.dilation
    padding = conv.padding
    if isinstance(padding, str):
        pads = []
        for k, d in ((k1, d1), (k2, d2)):
            if padding == "valid":
                left = right = 0
            else:
                total = d * (k - 1)
                left, right = total // 2, total - total // 2
            if left != right:
                raise UnsupportableModuleError("Unequal padding not supported in unfold.")
            pads.append(left)
        p1, p2 = pads
    else:
        p1, p2 = padding
    return k1, k2, s1, s2, p1, p2, d1, d2


def im2col(x: torch.Tensor, conv: nn.Conv2d, append_ones: bool, out_dtype: torch.dtype = torch.float32,
           row_multiple: int = 1) -> torch.Tensor:
    """Patches ``[b, P, I']`` of module/conv2d.py:15-64 (+ ones column), via kf_im2col.  ``row_multiple > 1``: the patch rows as
    ONE matrix ``[b P rounded up, I']`` whose trailing rows are zero (whole k-tiles for the K-major covariance kernel)."""
    nat.require_device(x, "x")
    x = _contig(x)
    b, c, h, w = x.shape
    k1, k2, s1, s2, p1, p2, d1, d2 = conv_geometry(conv)
    o1 = (h + 2 * p1 - d1 * (k1 - 1) - 1) // s1 + 1
    o2 = (w + 2 * p2 - d2 * (k2 - 1) - 1) // s2 + 1
    ip = (c // conv.groups) * k1 * k2 + int(append_ones)
    n = b * o1 * o2
    if row_multiple > 1:
        out = torch.empty((-(-n // row_multiple) * row_multiple, ip), dtype=out_dtype, device=x.device)
        out[n:].zero_()
    else:
        out = torch.empty((b, o1 * o2, ip), dtype=out_dtype, device=x.device)
    nat.check(
        nat.lib().kf_im2col(out.data_ptr(), nat.dtype_code(out_dtype), x.data_ptr(), nat.dtype_code(x.dtype), b, c, h, w,
                            k1, k2, s1, s2, p1, p2, d1, d2, conv.groups, int(append_ones), nat.stream_ptr(x.device)),
        "kf_im2col",
    )
    return out


def conv_patch_rows_cov(cov: torch.Tensor, count: torch.Tensor, x: torch.Tensor, conv: nn.Conv2d) -> bool:
    """A bf16 conv layer the implicit-im2col covariance does not take (an output grid that is not whole k-steps: ResNet-9's
    unpadded 3 x 3 layer with its 6 x 6 grid) on the K-MAJOR covariance kernel: the materialised patch rows ``[b P, I']`` ARE a
    K-major operand (kf_syrk_rows_bf16 on one "sample" of ``b P`` rows, zero rows up to a whole k-tile) -- 0.79 -> 0.36 ms per
    ResNet-9 batch against the generic kf_syrk_accum engine (round 6).  ``False``: not eligible (the caller falls back)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and conv.groups == 1):
        return False
    ip = x.shape[1] * conv.kernel_size[0] * conv.kernel_size[1] + int(conv.bias is not None)
    if ip % 8 != 0 or ip < 256 or ip >= COV_STAGED_MAX_DIM:
        return False
    rows = im2col(x, conv, conv.bias is not None, torch.bfloat16, row_multiple=64)
    k1, k2, s1, s2, p1, p2, d1, d2 = conv_geometry(conv)
    n = x.shape[0] * ((x.shape[2] + 2 * p1 - d1 * (k1 - 1) - 1) // s1 + 1) * ((x.shape[3] + 2 * p2 - d2 * (k2 - 1) - 1) // s2 + 1)
    if n <= 0 or not _syrk_rows_bf16(cov, rows.unsqueeze(0), None, False, 1.0):
        return False
    count.add_(n)
    return True


def conv2d_cov_geometry(x: torch.Tensor, conv: nn.Conv2d):
    """Arguments of ``kf_conv2d_cov_accum`` for this input, or ``None`` when the implicit-im2col covariance does not apply
    (not bf16 on the GPU, groups, bias, or an output grid that is not whole 16-byte chunks / 64-wide k-steps)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4) or conv.groups != 1 or conv.bias is not None:
        return None
    if not 0 < x.shape[0] <= 65535:
        return None
    geometry = tuple(x.shape) + conv_geometry(conv)
    return geometry if nat.lib().kf_conv2d_cov_workspace_bytes(*geometry) > 0 else None


def conv2d_cov_accum(cov: torch.Tensor, count: torch.Tensor, x: torch.Tensor, conv: nn.Conv2d, geometry) -> None:
    """``cov += patches^T patches`` without the patch tensor (kf_conv2d_cov_accum); ``count += b * O1 * O2``."""
    x = _contig(x)
    b, c, h, w, k1, k2, s1, s2, p1, p2, d1, d2 = geometry
    o1 = (h + 2 * p1 - d1 * (k1 - 1) - 1) // s1 + 1
    o2 = (w + 2 * p2 - d2 * (k2 - 1) - 1) // s2 + 1
    n, d = b * o1 * o2, c * k1 * k2
    ws_bytes = nat.lib().kf_conv2d_cov_workspace_bytes(*geometry)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    with _Timed("syrk_accum", x.device, float(n) * d * (d + 1), float(b) * c * h * w * 2):
        nat.check(
            nat.lib().kf_conv2d_cov_accum(cov.data_ptr(), cov.shape[1], x.data_ptr(), *geometry, 1.0, ws.data_ptr(), ws_bytes,
                                          nat.stream_ptr(x.device)),
            "kf_conv2d_cov_accum",
        )
    count.add_(n)


def conv2d_cov_small(cov: torch.Tensor, count: torch.Tensor, x: torch.Tensor, conv: nn.Conv2d) -> bool:
    """Patch width ``C k1 k2 (+ 1) <= 32`` (the first layer of an image model): covariance straight from the NCHW input of any
    float dtype on one fp32 MFMA per two positions (kf_conv2d_cov_small); ``False`` when the layer is not eligible."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16, torch.float16)) or conv.groups != 1:
        return False
    ones = conv.bias is not None
    b, c, h, w = x.shape
    k1, k2, s1, s2, p1, p2, d1, d2 = conv_geometry(conv)
    d = c * k1 * k2
    if d + int(ones) > 32 or os.environ.get("KF_CONV_COV_SMALL", "1") == "0":
        return False
    o1 = (h + 2 * p1 - d1 * (k1 - 1) - 1) // s1 + 1
    o2 = (w + 2 * p2 - d2 * (k2 - 1) - 1) // s2 + 1
    if o1 <= 0 or o2 <= 0:
        return False
    x = _contig(x, align=False)
    n = b * o1 * o2
    with _Timed("syrk_accum_f32" if x.dtype == torch.float32 else "syrk_accum", x.device, float(n) * (d + ones) * (d + ones + 1),
                float(x.numel()) * x.element_size()):
        nat.check(
            nat.lib().kf_conv2d_cov_small(cov.data_ptr(), cov.shape[1], x.data_ptr(), nat.dtype_code(x.dtype), b, c, h, w,
                                          k1, k2, s1, s2, p1, p2, d1, d2, int(ones), 1.0, nat.stream_ptr(x.device)),
            "kf_conv2d_cov_small",
        )
    count.add_(n)
    return True


def conv_activation_cov(cov: torch.Tensor, count: torch.Tensor, x: torch.Tensor, conv: nn.Conv2d) -> None:
    """module/conv2d.py:106-128 + tracker/factor.py:58."""
    geometry = conv2d_cov_geometry(x, conv)
    if geometry is not None:
        conv2d_cov_accum(cov, count, x, conv, geometry)
        return
    if conv2d_cov_small(cov, count, x, conv) or conv_patch_rows_cov(cov, count, x, conv):
        return
    patches = im2col(x, conv, conv.bias is not None, x.dtype if x.dtype != torch.float64 else torch.float32)
    n, d = patches.shape[0] * patches.shape[1], patches.shape[2]
    syrk_accum(cov, patches, n, d, max(n, 1), 0, d, 1, None, False, 1.0, count)


def conv_gradient_cov(cov: torch.Tensor, count: torch.Tensor, g: torch.Tensor, alpha: float = 1.0) -> None:
    """module/conv2d.py:130-132 (``b c o1 o2 -> (b o1 o2) c``) + tracker/factor.py:93, without the transpose copy."""
    g = _contig(g)
    b, o, h, w = g.shape
    p = h * w
    if (g.is_cuda and g.dtype == torch.bfloat16 and p % 64 == 0 and 0 < b <= 65535 and g.data_ptr() % 16 == 0
            and o < COV_STAGED_MAX_DIM):
        # the NCHW gradient IS the operand layout of the covariance kernel: C_out rows of O1*O2 contiguous values per sample
        ws_bytes = nat.lib().kf_syrk_planes_workspace_bytes(o)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
        with _Timed("syrk_accum", g.device, float(b * p) * o * (o + 1), float(b * p) * o * 2):
            nat.check(
                nat.lib().kf_syrk_planes_bf16(cov.data_ptr(), cov.shape[1], g.data_ptr(), b, o, p, alpha, ws.data_ptr(), ws_bytes,
                                              nat.stream_ptr(g.device)),
                "kf_syrk_planes_bf16",
            )
        count.add_(b * p)
        return
    if g.dtype == torch.bfloat16 and o % 8 == 0 and o > 8:
        # [b,P,O] rows feed the bf16 MFMA engine (k = row index strided, columns contiguous)
        rows = g.flatten(2).transpose(1, 2).contiguous()
        syrk_accum(cov, rows, b * p, o, max(b * p, 1), 0, o, 1, None, False, alpha, count)
        return
    syrk_accum(cov, g, b * p, o, p, o * p, 1, p, None, False, alpha, count)


# ---------------------------------------------------------------------------------------------
# GEMM building block
# ---------------------------------------------------------------------------------------------
def gemm(c: torch.Tensor, ldc: int, c_batch_stride: int, a: kf_view, b: kf_view, batch: int = 1, alpha: float = 1.0,
         beta: float = 0.0, mul: Optional[torch.Tensor] = None) -> None:
    nat.require_device(c, "c")
    _require(c.dtype == torch.float32, 'c.dtype == torch.float32')
    nat.check(
        nat.lib().kf_gemm(c.data_ptr(), ldc, c_batch_stride, ctypes.byref(a), ctypes.byref(b), batch, alpha, beta,
                          _ptr(mul), mul.shape[-1] if mul is not None else 0, nat.stream_ptr(c.device)),
        "kf_gemm",
    )


def matmul_nn(x: torch.Tensor, w: torch.Tensor, append_ones: bool = False) -> torch.Tensor:
    """``[x, 1] @ w`` for ``x: [n, d]`` (any float dtype), ``w: [d', m]`` fp32 -> fp32 ``[n, m]``."""
    x, w = _contig(x), _contig(w)
    n, d = x.shape
    out = torch.empty((n, w.shape[1]), dtype=torch.float32, device=x.device)
    gemm(out, w.shape[1], 0, view(x, 0, d, 1, n, d, ones_k=append_ones), view(w, 0, 1, w.shape[1], w.shape[1], w.shape[0]))
    return out


def rotate_bf16(x: torch.Tensor, q_t: torch.Tensor, bias_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ q (+ bias_row)`` on the bf16 MFMA engine: ``x: [n, d]`` bf16, ``q_t``: contiguous bf16 ``[m, ld]`` holding
    ``q^T`` in its first ``d`` columns (``ld >= d``, both multiples of 8) -> bf16 ``[n, m]`` -- the eigenbasis rotation of
    tracker/factor.py:218-226 in the reference's bf16 lambda_dtype.  ``bias_row`` (fp32, ``<= m`` entries) is added to
    every output row: with ``q_t`` built from the first ``I`` rows of ``Q_A`` and ``bias_row = Q_A[I]`` this is
    ``[x, 1] @ Q_A`` without materialising the ones column; zero rows of ``q_t`` give zero (padding) output columns."""
    x, q_t = _contig(x), _contig(q_t)
    n, d = x.shape
    m, ld = q_t.shape
    _require(ld >= d and x.dtype == q_t.dtype == torch.bfloat16, 'ld >= d and x.dtype == q_t.dtype == torch.bfloat16')
    out = torch.empty((n, m), dtype=torch.bfloat16, device=x.device)
    nat.require_device(x, "x")
    if bias_row is None:
        nat.check(
            nat.lib().kf_gemm_out(out.data_ptr(), nat.dtype_code(out.dtype), m, 0, ctypes.byref(view(x, 0, d, 1, n, d)),
                                  ctypes.byref(view(q_t, 0, ld, 1, m, d)), 1, 1.0, nat.stream_ptr(x.device)),
            "kf_gemm_out",
        )
        return out
    bias_row = _contig(bias_row)
    _require(bias_row.dtype == torch.float32 and bias_row.numel() <= m, 'bias_row.dtype == torch.float32 and bias_row.numel() <= m')
    nat.check(
        nat.lib().kf_gemm_bias_out(out.data_ptr(), m, ctypes.byref(view(x, 0, d, 1, n, d)), ctypes.byref(view(q_t, 0, ld, 1, m, d)),
                                   bias_row.data_ptr(), bias_row.numel(), nat.stream_ptr(x.device)),
        "kf_gemm_bias_out",
    )
    return out


def per_sample_gradient(g: torch.Tensor, a: torch.Tensor, append_ones: bool) -> torch.Tensor:
    """``einsum("b...i,b...o->bio", g, [a,1])`` of module/linear.py:72 / conv2d.py:176: ``g: [b,R,O]``, ``a: [b,R,I]``."""
    g, a = _contig(g), _contig(a)
    b, r, o = g.shape
    i = a.shape[2]
    ip = i + int(append_ones)
    out = torch.empty((b, o, ip), dtype=torch.float32, device=g.device)
    gemm(out, ip, o * ip, view(g, r * o, 1, o, o, r), view(a, r * i, 1, i, i, r, ones_row=append_ones), batch=b)
    return out


# ---------------------------------------------------------------------------------------------
# Stage 2: eigendecomposition, Lambda
# ---------------------------------------------------------------------------------------------
def eigh_stats(reset: bool = False) -> dict:
    """Paths ``eigh`` took for ``d >= 256`` since the last reset (kf_eigh_stats): factor-first solves, fall-backs to the
    solver that carries V, Cholesky retries at a larger shift."""
    first, fallback, retries = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    nat.lib().kf_eigh_stats(ctypes.byref(first), ctypes.byref(fallback), ctypes.byref(retries), int(reset))
    return {"factor_first": first.value, "fallback": fallback.value, "cholesky_retries": retries.value}


# relative rounding noise of a covariance by the dtype it was STORED in (``eigh(noise_rel=...)``): sizes the shift of the
# factor-first eigensolver (kf_eigh_f64).  fp32 / fp64: the library's defaults.
STORAGE_NOISE = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -10}


def eigh(cov: torch.Tensor, count: float, max_sweeps: int = 0, noise_rel: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """fp64 ``eigh(0.5 (cov + cov^T) / count)`` (factor/eigen.py:193-205) -> (evals, evecs, sweeps)."""
    nat.require_device(cov, "cov")
    _require(cov.dim() == 2 and cov.shape[0] == cov.shape[1] and cov.dtype in (torch.float32, torch.float64), 'cov.dim() == 2 and cov.shape[0] == cov.shape[1] and cov.dtype in (torch.float32, torch.float64)')
    cov = _contig(cov)
    d = cov.shape[0]
    evals = torch.empty(d, dtype=torch.float64, device=cov.device)
    evecs = torch.empty((d, d), dtype=torch.float64, device=cov.device)
    ws_bytes = nat.lib().kf_eigh_workspace_bytes(d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=cov.device)
    sweeps = ctypes.c_int(0)
    nat.check(
        nat.lib().kf_eigh_f64(cov.data_ptr(), nat.dtype_code(cov.dtype), float(count), float(noise_rel), d, evals.data_ptr(), evecs.data_ptr(),
                              ws.data_ptr(), ws_bytes, max_sweeps, ctypes.byref(sweeps), nat.stream_ptr(cov.device)),
        "kf_eigh_f64",
    )
    return evals, evecs, sweeps.value


EIGH_SMALL_MAX = 96


_EIGH_SMALL_SLOW_WARNED = False


def eigh_small(g: torch.Tensor, inv_sqrt: bool = False, floor_rel: float = 1e-12) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched eigendecomposition of ``[batch, l, l]`` fp32 symmetric matrices (kf_eigh_small_batched for ``l <= 96``, one
    ``kf_eigh_f64`` problem per matrix above):
    eigenvalues DESCENDING, eigenvectors in columns; ``inv_sqrt`` scales column j by ``1/sqrt(lambda_j)``."""
    nat.require_device(g, "g")
    g = _contig(g)
    batch, l, l2 = g.shape
    _require(g.dtype == torch.float32 and l == l2, 'g.dtype == torch.float32 and l == l2')
    if l > EIGH_SMALL_MAX:
        # beyond the in-LDS solver (ranks above 88): one kf_eigh_f64 problem per matrix -- slower (a host read-back per sweep
        # below d = 256), same result
        global _EIGH_SMALL_SLOW_WARNED
        if not _EIGH_SMALL_SLOW_WARNED:
            _EIGH_SMALL_SLOW_WARNED = True
            logging.getLogger("kronfluence_amd").warning(
                "query_gradient_low_rank above %d: the %d x %d Gram eigenproblems of the range finder are solved one at a time "
                "(kf_eigh_f64) instead of batched in LDS -- expect a slow query stage.", EIGH_SMALL_MAX - 8, l, l)
        evals = torch.empty((batch, l), dtype=torch.float32, device=g.device)
        evecs = torch.empty((batch, l, l), dtype=torch.float32, device=g.device)
        for j in range(batch):
            lam, vec, _ = eigh(g[j], 1.0)
            lam, vec = lam.flip(0), vec.flip(1)
            if inv_sqrt:
                clipped = torch.maximum(lam, floor_rel * lam[:1].clamp(min=0.0))
                vec = vec * torch.where(clipped > 0, clipped.rsqrt(), torch.zeros_like(clipped))
            evals[j], evecs[j] = lam.float(), vec.float()
        return evals, evecs
    evals = torch.empty((batch, l), dtype=torch.float32, device=g.device)
    evecs = torch.empty((batch, l, l), dtype=torch.float32, device=g.device)
    nat.check(
        nat.lib().kf_eigh_small_batched(g.data_ptr(), batch, l, evals.data_ptr(), evecs.data_ptr(), int(inv_sqrt), floor_rel, 0,
                                        nat.stream_ptr(g.device)),
        "kf_eigh_small_batched",
    )
    return evals, evecs


def _bmm(out_shape, a: kf_view, b: kf_view, batch: int, device, alpha: float = 1.0) -> torch.Tensor:
    out = torch.empty(out_shape, dtype=torch.float32, device=device)
    gemm(out, out_shape[2], out_shape[1] * out_shape[2], a, b, batch=batch, alpha=alpha)
    return out


def low_rank_factors(p: torch.Tensor, rank: int, power_iterations: int = 2, oversample: int = 8
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rank-``rank`` factors ``(left [q,O,k], right [q,k,I'])`` with ``left @ right ~= p`` for ``p: [q,O,I']`` fp32 --
    the truncated SVD of the reference's low-rank query batching (module/tracker/precondition.py:19-75), computed as
    a randomised range finder with subspace iteration (Halko et al. 2011, the algorithm behind ``torch.svd_lowrank``)
    entirely out of batched MFMA GEMMs and the in-LDS small eigensolver:

        Y = P Omega;  Q = orth(Y);  repeat: Q = orth(P orth(P^T Q));  B = Q^T P;  B B^T = W S^2 W^T;
        left = Q W_k,  right = W_k^T B            (left @ right = Q W_k W_k^T Q^T P)

    ``orth(Y) = Y V S^-1`` from the eigendecomposition of the ``l x l`` Gram matrix ``Y^T Y`` (``l = rank + oversample``,
    capped by ``min(O, I')`` -- where the range finder, hence the truncated SVD, is exact)."""
    p = _contig(p)
    _require(p.dtype == torch.float32 and p.dim() == 3, 'p.dtype == torch.float32 and p.dim() == 3')
    q, o, ip = p.shape
    l = min(rank + oversample, o, ip)
    k = min(rank, l)
    dev = p.device
    gen = torch.Generator(device=dev).manual_seed(0x5eed)
    omega_t = torch.randn((l, ip), generator=gen, dtype=torch.float32, device=dev)  # Omega^T, shared by the batch

    def orth(y: torch.Tensor, rows: int) -> torch.Tensor:  # y: [q, rows, l]
        gram = _bmm((q, l, l), view(y, rows * l, 1, l, l, rows), view(y, rows * l, 1, l, l, rows), q, dev)
        _, basis = eigh_small(gram, inv_sqrt=True, floor_rel=1e-10)
        return _bmm((q, rows, l), view(y, rows * l, l, 1, rows, l), view(basis, l * l, 1, l, l, l), q, dev)

    def p_times(z: torch.Tensor) -> torch.Tensor:  # [q,O,I'] x [q,I',l] -> [q,O,l]
        return _bmm((q, o, l), view(p, o * ip, ip, 1, o, ip), view(z, ip * l, 1, l, l, ip), q, dev)

    def pt_times(y: torch.Tensor) -> torch.Tensor:  # [q,O,I']^T x [q,O,l] -> [q,I',l]
        return _bmm((q, ip, l), view(p, o * ip, 1, ip, ip, o), view(y, o * l, 1, l, l, o), q, dev)

    basis = orth(_bmm((q, o, l), view(p, o * ip, ip, 1, o, ip), view(omega_t, 0, ip, 1, l, ip), q, dev), o)
    for _ in range(power_iterations):
        basis = orth(p_times(orth(pt_times(basis), ip)), o)
    b = _bmm((q, l, ip), view(basis, o * l, 1, l, l, o), view(p, o * ip, 1, ip, ip, o), q, dev)  # Q^T P
    small = _bmm((q, l, l), view(b, l * ip, ip, 1, l, ip), view(b, l * ip, ip, 1, l, ip), q, dev)  # B B^T
    _, w = eigh_small(small)
    left = _bmm((q, o, k), view(basis, o * l, l, 1, o, l), view(w, l * l, 1, l, k, l), q, dev)
    right = _bmm((q, k, ip), view(w, l * l, 1, l, k, l), view(b, l * ip, 1, ip, ip, l), q, dev)
    return left, right


def low_rank_product(left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
    """``left @ right`` -> fp32 ``[q, O, I']`` (reconstruction of low-rank query gradients ahead of the score GEMM)."""
    left, right = _contig(left), _contig(right)
    if left.dtype != torch.float32:
        left = cast(left, torch.float32)
    if right.dtype != torch.float32:
        right = cast(right, torch.float32)
    q, o, k = left.shape
    ip = right.shape[2]
    return _bmm((q, o, ip), view(left, o * k, k, 1, o, k), view(right, k * ip, 1, ip, ip, k), q, left.device)


def lambda_accum(lam: torch.Tensor, gt: torch.Tensor, at: torch.Tensor, b: int, r: int, scale: float = 1.0) -> None:
    """``lam += sum_b (Gt_b^T At_b)^2`` with rotated factors (kf_lambda_accum; tracker/factor.py:218-226).  ``at`` may be
    wider than ``lam`` (bf16 rows zero-padded to a multiple of 8, see ``rotate_bf16``)."""
    nat.require_device(lam, "lam")
    _require(lam.dtype == torch.float32 and gt.dtype == at.dtype and gt.dtype in (torch.float32, torch.bfloat16), 'lam.dtype == torch.float32 and gt.dtype == at.dtype and gt.dtype in (torch.float32, torch.bfloat16)')
    _require(gt.is_contiguous() and at.is_contiguous(), 'gt.is_contiguous() and at.is_contiguous()')
    o, ip = lam.shape
    ld_at = at.shape[-1]
    _require(gt.numel() == b * r * o and at.numel() == b * r * ld_at and ld_at >= ip, 'gt.numel() == b * r * o and at.numel() == b * r * ld_at and ld_at >= ip')
    # the product of the rotated factors, squared and summed (the rotations themselves are kf_gemm calls)
    with _Timed("lambda_accum", lam.device, 2.0 * b * r * o * ip, float(b) * r * (o + ip) * gt.element_size()):
        nat.check(
            nat.lib().kf_lambda_accum(lam.data_ptr(), ip, gt.data_ptr(), at.data_ptr(), ld_at, nat.dtype_code(gt.dtype), b, r, o, ip,
                                      scale, nat.stream_ptr(lam.device)),
            "kf_lambda_accum",
        )


def rotate_rows_transposed(x: torch.Tensor, q_t: torch.Tensor, bias_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[s, j, r] = sum_k q_t[j, k] x[s, r, k] (+ bias_row[j])`` for ``x: [n, R, d]`` bf16 and ``q_t``: contiguous bf16
    ``[m, ld]`` holding the transposed eigenvector matrix in its first ``d`` columns -> bf16 ``[n, m, R]``: the eigenbasis
    rotation of ``rotate_bf16`` written K-contiguous per sample, the operand layout of ``lambda_rows_accum``
    (kf_rotate_rows_transposed_bf16)."""
    nat.require_device(x, "x")
    x, q_t = _contig(x), _contig(q_t)
    n, r, d = x.shape
    m, ld = q_t.shape
    _require(x.dtype == q_t.dtype == torch.bfloat16 and ld >= d, "rotate_rows_transposed: bf16 operands, ld >= d")
    if bias_row is not None:
        bias_row = _contig(bias_row)
        _require(bias_row.dtype == torch.float32 and bias_row.numel() <= m, "rotate_rows_transposed: fp32 bias of <= m entries")
    out = torch.empty((n, m, r), dtype=torch.bfloat16, device=x.device)
    nat.check(
        nat.lib().kf_rotate_rows_transposed_bf16(out.data_ptr(), x.data_ptr(), n, r, d, q_t.data_ptr(), ld, m, _ptr(bias_row),
                                                 bias_row.numel() if bias_row is not None else 0, nat.stream_ptr(x.device)),
        "kf_rotate_rows_transposed_bf16",
    )
    return out


def lambda_rows_eligible(o: int, i: int, r: int) -> bool:
    """Shapes ``lambda_rows_accum`` (and the rotations that feed it) take: whole 64-deep k-tiles in all three contractions."""
    return o % 64 == 0 and i % 64 == 0 and r % 64 == 0 and o >= 128 and i >= 64


def lambda_rows_accum(lam: torch.Tensor, gt_t: torch.Tensor, at_t: torch.Tensor, scale: float = 1.0) -> None:
    """``lam[o, i] += scale^2 * sum_s (sum_r gt_t[s, o, r] at_t[s, i, r])^2`` (kf_lambda_rows_accum): the Lambda update of
    tracker/factor.py:218-226 from K-contiguous rotated factors ``gt_t: [b, O, R]``, ``at_t: [b, W, R]`` (``W >= I'``)."""
    nat.require_device(lam, "lam")
    nat.require_device(gt_t, "gt_t")
    _require(lam.dtype == torch.float32 and lam.is_contiguous() and gt_t.dtype == at_t.dtype == torch.bfloat16
             and gt_t.is_contiguous() and at_t.is_contiguous(), "lambda_rows_accum: fp32 Lambda, contiguous bf16 factors")
    o, ip = lam.shape
    b, o2, r = gt_t.shape
    w = at_t.shape[1]
    _require(o2 == o and at_t.shape[0] == b and at_t.shape[2] == r and w >= ip, "lambda_rows_accum: shapes")
    with _Timed("lambda_accum", lam.device, 2.0 * b * r * o * ip, float(b) * r * (o + ip) * 2):
        nat.check(
            nat.lib().kf_lambda_rows_accum(lam.data_ptr(), ip, gt_t.data_ptr(), at_t.data_ptr(), b, r, o, w, ip, scale,
                                           nat.stream_ptr(lam.device)),
            "kf_lambda_rows_accum",
        )


def rotate_channels(g_nchw: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """``out[n, o', p] = sum_o q[o, o'] g[n, o, p]`` for an NCHW bf16 tensor and ``q``: bf16 ``[O, O]`` -- the gradient-side
    eigenbasis rotation applied along the channel axis, result in the same NCHW layout (batched TN product on the bf16 engine)."""
    nat.require_device(g_nchw, "g_nchw")
    g_nchw, q = _contig(g_nchw), _contig(q)
    b, o = g_nchw.shape[0], g_nchw.shape[1]
    p = g_nchw.numel() // (b * o)
    _require(q.shape == (o, o) and q.dtype == g_nchw.dtype == torch.bfloat16 and o % 8 == 0 and p % 8 == 0,
             "rotate_channels: bf16 operands, O and the positions multiples of 8")
    out = torch.empty_like(g_nchw)
    nat.check(
        nat.lib().kf_gemm_out(out.data_ptr(), nat.dtype_code(out.dtype), p, o * p, ctypes.byref(view(q, 0, 1, o, o, o)),
                              ctypes.byref(view(g_nchw, o * p, 1, p, p, o)), b, 1.0, nat.stream_ptr(g_nchw.device)),
        "kf_gemm_out",
    )
    return out


def lambda_conv2d_geometry(x_shape, out_channels: int, conv: nn.Conv2d):
    """Geometry tuple of the dense-form Lambda call, or ``None`` when the layer / batch is not eligible (decided by the library:
    ``kf_lambda_conv2d_workspace_bytes`` returns -1)."""
    if conv.groups != 1 or conv.bias is not None:
        return None
    b, c, h, w = x_shape
    geometry = (b, c, h, w, out_channels) + conv_geometry(conv)
    return geometry if nat.lib().kf_lambda_conv2d_workspace_bytes(*geometry) > 0 else None


def lambda_conv2d_accum(lam: torch.Tensor, gt_nchw: torch.Tensor, x: torch.Tensor, geometry, qa_t_perm: torch.Tensor,
                        scale: float = 1.0) -> None:
    """``lam += scale^2 * sum_n (Qg^T g_n Qa)^2`` of a Conv2d layer in the dense form with implicit im2col
    (kf_lambda_conv2d_accum; tracker/factor.py:218-226 + module/conv2d.py:164-177).  ``gt_nchw``: the output gradient already
    rotated along its channel axis (``rotate_channels``); ``qa_t_perm``: see ``conv_patch_order_eigenvectors``."""
    nat.require_device(lam, "lam")
    gt_nchw, x, qa_t_perm = _contig(gt_nchw), _contig(x), _contig(qa_t_perm)
    _require(lam.dtype == torch.float32 and lam.is_contiguous() and gt_nchw.dtype == x.dtype == qa_t_perm.dtype == torch.bfloat16,
             "lambda_conv2d_accum: fp32 Lambda, bf16 operands")
    o, ip = lam.shape
    b, c, h, w = x.shape
    k1, k2 = geometry[5], geometry[6]
    r = gt_nchw.shape[2] * gt_nchw.shape[3]
    _require(gt_nchw.shape[:2] == (b, o) and ip == c * k1 * k2 and qa_t_perm.shape[0] >= ip, "lambda_conv2d_accum: shapes")
    ws_bytes = nat.lib().kf_lambda_conv2d_workspace_bytes(*geometry)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    flops = 2.0 * b * r * o * ip + 2.0 * b * o * ip * ip
    with _Timed("lambda_accum", lam.device, flops, float(b) * (r * o + c * h * w) * 2):
        nat.check(
            nat.lib().kf_lambda_conv2d_accum(lam.data_ptr(), lam.shape[1], gt_nchw.data_ptr(), x.data_ptr(), *geometry,
                                             qa_t_perm.data_ptr(), ip, qa_t_perm.shape[1], scale, ws.data_ptr(), ws_bytes,
                                             nat.stream_ptr(x.device)),
            "kf_lambda_conv2d_accum",
        )


def lambda_conv2d_channels(geometry) -> int:
    """Padded channel count of the dense-form Lambda call for this geometry (kf_lambda_conv2d_channels): what
    ``conv_patch_order_eigenvectors`` must lay ``Q_A^T`` out with."""
    return int(nat.lib().kf_lambda_conv2d_channels(*geometry))


def conv_patch_order_eigenvectors(q_a: torch.Tensor, channels: int, taps: int, padded_channels: Optional[int] = None) -> torch.Tensor:
    """``Q_A^T`` for the implicit-im2col kernels: bf16 ``[I' rounded up to 8, taps * Cp]`` whose row ``i'`` is column ``i'`` of
    ``q_a`` with its rows re-ordered from the reference's patch order ``(c, ky, kx)`` to the kernels' ``(ky, kx, c)``, ``c``
    zero-padded to ``Cp`` = ``padded_channels`` (default: the next multiple of 8)."""
    ip = q_a.shape[0]
    cp = padded_channels if padded_channels else channels + (-channels) % 8
    q = q_a.reshape(channels, taps, ip).transpose(0, 1)              # [taps, C, I']
    q = torch.nn.functional.pad(q, (0, 0, 0, cp - channels))          # [taps, Cp, I']
    out = q.reshape(taps * cp, ip).t()                               # [I', taps * Cp]
    return torch.nn.functional.pad(out, (0, 0, 0, (-ip) % 8)).to(torch.bfloat16).contiguous()


# ---------------------------------------------------------------------------------------------
# Stage 3: preconditioning and pairwise scores
# ---------------------------------------------------------------------------------------------
def inv_lambda(lam: torch.Tensor, n_lambda: float, damping: Optional[float]) -> torch.Tensor:
    """``1 / (lam / n + damping)`` in fp64 (factor/config.py:331-338); ``None`` -> 0.1 * mean heuristic."""
    nat.require_device(lam, "lam")
    lam = _contig(lam.to(torch.float32))
    out = torch.empty_like(lam)
    ws = torch.empty(2, dtype=torch.float64, device=lam.device)
    nat.check(
        nat.lib().kf_inv_lambda(out.data_ptr(), lam.data_ptr(), lam.numel(), float(n_lambda),
                                -1.0 if damping is None else float(damping), ws.data_ptr(), nat.stream_ptr(lam.device)),
        "kf_inv_lambda",
    )
    return out


def precondition(g: torch.Tensor, a: torch.Tensor, append_ones: bool, q_g: torch.Tensor, q_a: torch.Tensor,
                 lam_inv: torch.Tensor, scale: float = 1.0, out_dtype: torch.dtype = torch.float32,
                 q_a_bf16: Optional[torch.Tensor] = None, q_g_t_bf16: Optional[torch.Tensor] = None,
                 q_a_t_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """EK-FAC preconditioned per-sample gradient from ``g: [q,R,O]`` and ``a: [q,R,I]`` (tracker/precondition.py:102-123 +
    factor/config.py:341-353): ``[q, O, I']``.  With the bf16 eigenvector copies (``[W, W]``, ``W = I'`` rounded up to a
    multiple of 8, zero-padded), bf16 factors and several rows per sample, everything runs on the bf16 MFMA engine and the
    result is ``[q, O, W]`` -- ``W - I'`` trailing zero columns (none unless ``I'`` is odd)."""
    nat.require_device(g, "g")
    g, a = _contig(g), _contig(a)
    q, r, o = g.shape
    i = a.shape[2]
    ip = i + int(append_ones)
    _require(q_g.shape == (o, o) and q_a.shape == (ip, ip) and lam_inv.shape == (o, ip), 'q_g.shape == (o, o) and q_a.shape == (ip, ip) and lam_inv.shape == (o, ip)')
    _require(q_g.dtype == q_a.dtype == lam_inv.dtype == torch.float32 and g.dtype == a.dtype, 'q_g.dtype == q_a.dtype == lam_inv.dtype == torch.float32 and g.dtype == a.dtype')
    q_g, q_a, lam_inv = _contig(q_g), _contig(q_a), _contig(lam_inv)  # keep the contiguous copies alive
    _require(out_dtype in (torch.float32, torch.bfloat16), 'out_dtype in (torch.float32, torch.bfloat16)')
    width, ldq = ip, ip
    low = (q_a_bf16 is not None and q_g_t_bf16 is not None and q_a_t_bf16 is not None and out_dtype == torch.bfloat16
           and g.dtype == torch.bfloat16 and r > 1 and o % 8 == 0 and i % 8 == 0 and i >= 64 and o >= 64)
    if low:
        ldq = q_a_bf16.shape[0]
        _require(q_a_bf16.shape == q_a_t_bf16.shape == (ldq, ldq) and ldq % 8 == 0 and ip <= ldq < ip + 8, 'q_a_bf16.shape == q_a_t_bf16.shape == (ldq, ldq) and ldq % 8 == 0 and ip <= ldq < ip + 8')
        width = ldq
    else:
        q_a_bf16 = q_g_t_bf16 = q_a_t_bf16 = None
    out = torch.empty((q, o, width), dtype=out_dtype, device=g.device)
    ws_bytes = nat.lib().kf_precondition_workspace_bytes(q, r, o, ip)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    flops = 2.0 * q * r * (o * o + i * ip) + (2.0 * q * r * o * ip if r > 1 else 0.0) + 2.0 * q * o * ip * (ip + o)
    with _Timed("precondition", g.device, flops, float(q) * (r * (o + i) * g.element_size() + o * width * out.element_size())):
        nat.check(
            nat.lib().kf_precondition(out.data_ptr(), nat.dtype_code(out_dtype), width, g.data_ptr(), a.data_ptr(),
                                      nat.dtype_code(g.dtype), q, r, o, i,
                                      int(append_ones), q_g.data_ptr(), q_a.data_ptr(), lam_inv.data_ptr(), scale,
                                      _ptr(q_a_bf16), _ptr(q_g_t_bf16), _ptr(q_a_t_bf16), ldq, ws.data_ptr(), ws_bytes,
                                      nat.stream_ptr(g.device)),
            "kf_precondition",
        )
    return out


def precondition_bf16_eligible(g: torch.Tensor, a: torch.Tensor) -> bool:
    """Shapes / dtypes the bf16 form of the preconditioner takes (``precondition_bf16``; the library decides the same way)."""
    return (g.is_cuda and g.dtype == a.dtype == torch.bfloat16 and g.dim() == 3 and g.shape[1] > 1 and g.shape[2] % 8 == 0
            and a.shape[2] % 8 == 0 and g.shape[2] >= 64 and a.shape[2] >= 64)


def precondition_bf16(g: torch.Tensor, a: torch.Tensor, append_ones: bool, q_g_bf16: torch.Tensor, q_g_t_bf16: torch.Tensor,
                      q_a_bf16: torch.Tensor, q_a_t_bf16: torch.Tensor, bias_row: Optional[torch.Tensor], lam_inv: torch.Tensor,
                      scale: float = 1.0) -> torch.Tensor:
    """``precondition`` for ``precondition_dtype == score_dtype == bf16`` from bf16 eigenvectors ONLY (kf_precondition_bf16): no
    fp32 copies of ``Q_G`` / ``Q_A`` have to exist.  ``q_a_bf16`` / ``q_a_t_bf16``: ``[W, W]``, ``W = I'`` rounded up to a multiple of
    8, zero-padded; ``bias_row``: fp32 ``Q_A[I]`` (``append_ones``).  -> bf16 ``[q, O, W]``."""
    nat.require_device(g, "g")
    g, a = _contig(g), _contig(a)
    q, r, o = g.shape
    i = a.shape[2]
    ip = i + int(append_ones)
    w = q_a_bf16.shape[0]
    _require(precondition_bf16_eligible(g, a), "precondition_bf16: bf16 [q, R > 1, .] factors with O, I multiples of 8 and >= 64")
    _require(q_g_bf16.shape == q_g_t_bf16.shape == (o, o) and q_a_bf16.shape == q_a_t_bf16.shape == (w, w) and w % 8 == 0
             and ip <= w < ip + 8 and lam_inv.shape == (o, ip), "precondition_bf16: operand shapes")
    _require(q_g_bf16.dtype == q_g_t_bf16.dtype == q_a_bf16.dtype == q_a_t_bf16.dtype == torch.bfloat16
             and lam_inv.dtype == torch.float32 and all(t.is_contiguous() for t in (q_g_bf16, q_g_t_bf16, q_a_bf16, q_a_t_bf16, lam_inv)),
             "precondition_bf16: contiguous bf16 eigenvector copies, fp32 inverse Lambda")
    if append_ones:
        _require(bias_row is not None and bias_row.dtype == torch.float32 and bias_row.is_contiguous() and bias_row.numel() == ip,
                 "precondition_bf16: fp32 bias row of I' entries")
    out = torch.empty((q, o, w), dtype=torch.bfloat16, device=g.device)
    ws_bytes = nat.lib().kf_precondition_workspace_bytes(q, r, o, ip)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    flops = 2.0 * q * r * (o * o + i * ip) + 2.0 * q * r * o * ip + 2.0 * q * o * ip * (ip + o)
    with _Timed("precondition", g.device, flops, float(q) * (r * (o + i) * 2 + o * w * 2)):
        nat.check(
            nat.lib().kf_precondition_bf16(out.data_ptr(), w, g.data_ptr(), a.data_ptr(), q, r, o, i, int(append_ones),
                                           q_g_bf16.data_ptr(), q_g_t_bf16.data_ptr(), q_a_bf16.data_ptr(), q_a_t_bf16.data_ptr(), w,
                                           _ptr(bias_row) if append_ones else None, lam_inv.data_ptr(), scale, ws.data_ptr(), ws_bytes,
                                           nat.stream_ptr(g.device)),
            "kf_precondition_bf16",
        )
    return out


# When set to a dict, the instrumented entry points append ``(start_event, end_event, algorithmic_flops,
# algorithmic_bytes)`` under their name: bench.py times the hot kernels with HIP events on the launch stream.
EVENT_LOG: Optional[dict] = None


class _Timed:
    """``with _Timed(name, device, flops, bytes):`` brackets a C-ABI call with HIP events when ``EVENT_LOG`` is on."""

    def __init__(self, name: str, device, flops: float, nbytes: float) -> None:
        self.name, self.device, self.flops, self.nbytes = name, device, flops, nbytes
        self.start = None

    def __enter__(self):
        # (never inside a hipGraph capture: events recorded there cannot be timed -- the eager batches of a pass carry the timing)
        if EVENT_LOG is not None and not torch.cuda.is_current_stream_capturing():
            self.start, self.end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, kind, value, trace):
        if self.start is not None and kind is None and EVENT_LOG is not None:
            self.end.record(torch.cuda.current_stream(self.device))
            EVENT_LOG.setdefault(self.name, []).append((self.start, self.end, self.flops, self.nbytes))
        return False


def pairwise_score(scores: torch.Tensor, col_offset: int, p, g: torch.Tensor, a: torch.Tensor,
                   append_ones: bool, scale: float = 1.0) -> None:
    """``scores[:, col_offset:col_offset+b] += scale * <P_q, g_n>`` (kf_pairwise_score).

    ``scores``: fp32 ``[Q, N]`` device buffer shared by all layers and all train batches of a shard.
    ``p``: dense ``[Q, O, I']`` (fp32 / bf16), or an object with ``.tiled`` (bf16 k-tile-major ``[O*I'/64, Q, 64]``, see
    ``k_tile_major``) and ``.shape == (Q, O, I')`` -- ``R > 1`` only."""
    nat.require_device(scores, "scores")
    tiled = getattr(p, "tiled", None)
    storage = tiled if tiled is not None else p
    nat.require_device(storage, "p")
    _require(scores.dtype == torch.float32 and storage.dtype in (torch.float32, torch.bfloat16), 'scores.dtype == torch.float32 and storage.dtype in (torch.float32, torch.bfloat16)')
    _require(scores.is_contiguous() and storage.is_contiguous(), 'scores.is_contiguous() and storage.is_contiguous()')
    g, a = _contig(g), _contig(a)
    _require(g.dtype == a.dtype, 'g.dtype == a.dtype')
    b, r, o = g.shape
    i = a.shape[2]
    ip = i + int(append_ones)
    q = p.shape[0]
    _require(p.shape[1] == o and p.shape[2] == ip and scores.shape[0] == q and col_offset + b <= scores.shape[1], 'p.shape[1] == o and p.shape[2] == ip and scores.shape[0] == q and col_offset + b <= scores.shape[1]')
    ws_bytes = nat.lib().kf_pairwise_workspace_bytes(b, r, o, ip)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    flops = 2.0 * q * b * o * ip + (2.0 * b * r * o * ip if r > 1 else 0.0)
    # B_pair of SURVEY.md 8(d) for one call: the hooked factors once, P once, the score block read-modify-write
    nbytes = b * r * (o + i) * g.element_size() + q * o * ip * storage.element_size() + 2.0 * q * b * 4
    with _Timed("pairwise_score", g.device, flops, nbytes):
        nat.check(
            nat.lib().kf_pairwise_score(scores.data_ptr() + 4 * col_offset, scores.shape[1], storage.data_ptr(),
                                        nat.dtype_code(storage.dtype), q * 64 if tiled is not None else 0, q, g.data_ptr(),
                                        a.data_ptr(), nat.dtype_code(g.dtype), b, r, o, i, int(append_ones), scale,
                                        ws.data_ptr(), ws_bytes, nat.stream_ptr(g.device)),
            "kf_pairwise_score",
        )


def conv2d_score_geometry(x_shape, out_channels: int, conv: nn.Conv2d):
    """``(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2)`` of the implicit-im2col score call, or ``None`` when the layer is
    not eligible (groups, bias, or a geometry whose padding to whole 16-byte chunks / 64-wide k-steps would more than
    double the work -- decided by the library itself: ``kf_pairwise_conv2d_workspace_bytes`` returns -1)."""
    if conv.groups != 1 or conv.bias is not None:
        return None
    b, c, h, w = x_shape
    geometry = (b, c, h, w, out_channels) + conv_geometry(conv)
    return geometry if nat.lib().kf_pairwise_conv2d_workspace_bytes(*geometry) > 0 else None


def pairwise_score_conv2d(scores: torch.Tensor, col_offset: int, p, g_nchw: torch.Tensor, x: torch.Tensor, conv: nn.Conv2d,
                          scale: float = 1.0) -> None:
    """Implicit-im2col score of a Conv2d layer (kf_pairwise_score_conv2d): ``p.tiled`` is the k-tile-major bf16 P whose
    patch axis is ordered ``(ky, kx, c)`` with ``c`` zero-padded to a multiple of 8; ``g_nchw`` the hooked output gradient
    ``[b, O, O1, O2]`` and ``x`` the hooked input ``[b, C, H, W]``, both bf16 -- neither patches nor a transposed gradient
    are materialised."""
    nat.require_device(scores, "scores")
    nat.require_device(g_nchw, "g_nchw")
    g_nchw, x = _contig(g_nchw), _contig(x)
    _require(g_nchw.dtype == x.dtype == p.tiled.dtype == torch.bfloat16 and scores.dtype == torch.float32, 'g_nchw.dtype == x.dtype == p.tiled.dtype == torch.bfloat16 and scores.dtype == torch.float32')
    b, c, h, w = x.shape
    o, o1, o2 = g_nchw.shape[1], g_nchw.shape[2], g_nchw.shape[3]
    geometry = conv2d_score_geometry(x.shape, o, conv)
    _require(geometry is not None, "layer not eligible for the implicit-im2col path")
    k1, k2 = geometry[5], geometry[6]
    q, ip = p.shape[0], (c + (-c) % 8) * k1 * k2
    _require(p.shape == (q, o, ip) and scores.shape[0] == q and col_offset + b <= scores.shape[1], 'p.shape == (q, o, ip) and scores.shape[0] == q and col_offset + b <= scores.shape[1]')
    ws_bytes = nat.lib().kf_pairwise_conv2d_workspace_bytes(*geometry)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    r, real_ip = o1 * o2, c * k1 * k2
    flops = 2.0 * q * b * o * real_ip + 2.0 * b * r * o * real_ip
    nbytes = b * (r * o + c * h * w) * 2 + q * o * real_ip * 2 + 2.0 * q * b * 4  # B_pair: I^raw (not the patches), G, P, scores
    with _Timed("pairwise_score", x.device, flops, nbytes):
        nat.check(
            nat.lib().kf_pairwise_score_conv2d(scores.data_ptr() + 4 * col_offset, scores.shape[1], p.tiled.data_ptr(), q,
                                               g_nchw.data_ptr(), x.data_ptr(), *geometry, scale, ws.data_ptr(), ws_bytes,
                                               nat.stream_ptr(x.device)),
            "kf_pairwise_score_conv2d",
        )


def pairwise_score_rows(scores: torch.Tensor, col_offset: int, p, g: torch.Tensor, a: torch.Tensor, append_ones: bool,
                        scale: float = 1.0, second: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> None:
    """Score of a Linear layer on ``[b, R, .]`` activations on the LDS-DMA kernels (kf_pairwise_score_rows): ``p.tiled`` is
    k-tile-major bf16 with the augmented axis padded to ``p.shape[2]`` (a multiple of 8); bias column and padding of the
    train side are generated inside the call.  ``second = (g1, a1)``: a second train micro-batch whose scores go to the columns
    right behind the first one's -- both are contracted in ONE score GEMM (kf_pairwise_score_rows2)."""
    nat.require_device(scores, "scores")
    nat.require_device(g, "g")
    g, a = _contig(g), _contig(a)
    _require(g.dtype == a.dtype == p.tiled.dtype == torch.bfloat16 and scores.dtype == torch.float32, 'g.dtype == a.dtype == p.tiled.dtype == torch.bfloat16 and scores.dtype == torch.float32')
    b0, r, o = g.shape
    i = a.shape[2]
    g1 = a1 = None
    b1 = 0
    if second is not None:
        g1, a1 = _contig(second[0]), _contig(second[1])
        b1 = g1.shape[0]
        _require(g1.dtype == a1.dtype == torch.bfloat16 and tuple(g1.shape[1:]) == (r, o) and tuple(a1.shape) == (b1, r, i),
                 "pairwise_score_rows: the second segment must have the first one's row and feature counts")
    b = b0 + b1
    q, ipp = p.shape[0], p.shape[2]
    _require(p.shape[1] == o and scores.shape[0] == q and col_offset + b <= scores.shape[1], 'p.shape[1] == o and scores.shape[0] == q and col_offset + b <= scores.shape[1]')
    ws_bytes = nat.lib().kf_pairwise_rows_workspace_bytes(b, r, o, ipp)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    ip = i + int(append_ones)
    flops = 2.0 * q * b * o * ip + 2.0 * b * r * o * ip
    nbytes = b * r * (o + i) * 2 + q * o * ip * 2 + 2.0 * q * b * 4
    with _Timed("pairwise_score", g.device, flops, nbytes):
        nat.check(
            nat.lib().kf_pairwise_score_rows2(scores.data_ptr() + 4 * col_offset, scores.shape[1], p.tiled.data_ptr(), q,
                                              g.data_ptr(), a.data_ptr(), b0, _ptr(g1), _ptr(a1), b1, r, o, i, ipp, int(append_ones),
                                              scale, ws.data_ptr(), ws_bytes, nat.stream_ptr(g.device)),
            "kf_pairwise_score_rows2",
        )


def rowwise_dot(out: torch.Tensor, x: torch.Tensor, y: torch.Tensor, weight: Optional[torch.Tensor] = None,
                scale: float = 1.0, accumulate: bool = True) -> None:
    """``out[r] (+)= scale * sum_i x[r,i] y[r,i] (weight[i])`` (kf_rowwise_dot): the reduction of self-influence
    scores (module/tracker/self_score.py:61-62, 164).  ``x``, ``y``: ``[rows, ...]`` fp32 / bf16; ``out``: fp32 ``[rows]``."""
    nat.require_device(out, "out")
    nat.require_device(x, "x")
    nat.require_device(y, "y")
    x, y = _contig(x), _contig(y)
    rows = x.shape[0]
    d = x.numel() // max(rows, 1)
    _require(out.dtype == torch.float32 and out.is_contiguous() and out.numel() == rows and y.numel() == x.numel(), 'out.dtype == torch.float32 and out.is_contiguous() and out.numel() == rows and y.numel() == x.numel()')
    _require(x.dtype in (torch.float32, torch.bfloat16) and y.dtype in (torch.float32, torch.bfloat16), 'x.dtype in (torch.float32, torch.bfloat16) and y.dtype in (torch.float32, torch.bfloat16)')
    if weight is not None:
        nat.require_device(weight, "weight")
        weight = _contig(weight)
        _require(weight.dtype == torch.float32 and weight.numel() == d, 'weight.dtype == torch.float32 and weight.numel() == d')
    nat.check(
        nat.lib().kf_rowwise_dot(out.data_ptr(), x.data_ptr(), nat.dtype_code(x.dtype), y.data_ptr(), nat.dtype_code(y.dtype),
                                 _ptr(weight), rows, d, scale, int(accumulate), nat.stream_ptr(x.device)),
        "kf_rowwise_dot",
    )


def lowrank_rows_dot(scores: torch.Tensor, col_offset: int, u: torch.Tensor, v: torch.Tensor, b: int, r: int, q: int, k: int,
                     scale: float = 1.0) -> None:
    """``scores[j, col_offset + n] += scale * sum_{t, c} u[n r + t, j k + c] v[n r + t, j k + c]`` for bf16 ``u, v: [b r, q k]``
    (kf_lowrank_rows_dot): the reduction of the factored low-rank score of module/linear.py:83-99."""
    nat.require_device(scores, "scores")
    nat.require_device(u, "u")
    _require(scores.dtype == torch.float32 and scores.is_contiguous() and u.dtype == v.dtype == torch.bfloat16
             and u.is_contiguous() and v.is_contiguous() and u.shape == v.shape == (b * r, q * k),
             "lowrank_rows_dot: fp32 scores, contiguous bf16 [b r, q k] operands")
    _require(scores.shape[0] >= q and col_offset + b <= scores.shape[1], "lowrank_rows_dot: score block too small")
    nat.check(
        nat.lib().kf_lowrank_rows_dot(scores.data_ptr() + 4 * col_offset, scores.shape[1], u.data_ptr(), v.data_ptr(), b, r, q, k, scale,
                                      nat.stream_ptr(u.device)),
        "kf_lowrank_rows_dot",
    )


def mul_bcast(x: torch.Tensor, m: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """``scale * x[r] o m`` for ``x: [rows, ...]`` (fp32 / bf16) and ``m`` fp32 of one row's shape -> fp32 (kf_mul_bcast;
    the diagonal strategy's ``gradient * lambda_matrix``, factor/config.py:215-222)."""
    nat.require_device(x, "x")
    nat.require_device(m, "m")
    x, m = _contig(x), _contig(m)
    rows = x.shape[0]
    d = m.numel()
    _require(m.dtype == torch.float32 and x.numel() == rows * d and x.dtype in (torch.float32, torch.bfloat16), 'm.dtype == torch.float32 and x.numel() == rows * d and x.dtype in (torch.float32, torch.bfloat16)')
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    nat.check(
        nat.lib().kf_mul_bcast(out.data_ptr(), x.data_ptr(), nat.dtype_code(x.dtype), m.data_ptr(), rows, d, scale,
                               nat.stream_ptr(x.device)),
        "kf_mul_bcast",
    )
    return out


def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Device-side dtype conversion used when exporting fp32/fp64 accumulators in the factor dtype."""
    nat.require_device(src, "src")
    src = _contig(src)
    if src.dtype == dtype:
        return src.clone()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    nat.check(
        nat.lib().kf_cast(out.data_ptr(), nat.dtype_code(dtype), src.data_ptr(), nat.dtype_code(src.dtype), src.numel(),
                          nat.stream_ptr(src.device)),
        "kf_cast",
    )
    return out
