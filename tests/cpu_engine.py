"""TEST INFRASTRUCTURE ONLY: a torch-CPU stand-in for the leaf functions of ``kronfluence_amd.ops``.

The product has no CPU path (``ops`` hands raw device pointers to ``libkronfluence_hip.so`` and raises
without an MI355X).  To exercise the HOST logic on a GPU-less machine -- trackers, stage loops,
partitioning, aggregation, file layout, strategies -- the ``cpu_engine`` fixture (tests/conftest.py)
swaps the leaf operators for the functions below, which interpret the same arguments (including
``kf_view`` operand descriptions) with plain torch arithmetic in fp64, and lifts the Analyzer's
"needs a GPU" guard.  Nothing in ``kronfluence_amd/`` imports this module; the ``-m gpu`` tests run
the same scenarios through the real C ABI.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from oracle import ekfac_ref as ref


@dataclass
class View:
    t: torch.Tensor
    batch_stride: int
    row_stride: int
    k_stride: int
    rows: int
    depth: int
    ones_row: bool
    ones_k: bool
    square: bool
    k_tile_stride: int

    def dense(self, batch: int) -> torch.Tensor:
        """``[batch, rows(+1), depth(+1)]`` in fp64."""
        assert self.k_tile_stride == 0, "k-tile-major operands are a GPU-layout detail"
        flat = self.t.reshape(-1)
        x = flat.as_strided((batch, self.rows, self.depth), (self.batch_stride, self.row_stride, self.k_stride)).double()
        if self.square:
            x = x * x
        if self.ones_k:
            x = torch.cat([x, x.new_ones((batch, self.rows, 1))], dim=2)
        if self.ones_row:
            x = torch.cat([x, x.new_ones((batch, 1, x.shape[2]))], dim=1)
        return x


def view(t, batch_stride, row_stride, k_stride, rows, depth, ones_row=False, ones_k=False, square=False,
         k_tile_stride=0) -> View:
    assert t.is_contiguous()
    return View(t, batch_stride, row_stride, k_stride, rows, depth, bool(ones_row), bool(ones_k), bool(square), k_tile_stride)


def _strided_out(c: torch.Tensor, batch: int, m: int, n: int, ldc: int, c_batch_stride: int) -> torch.Tensor:
    return c.as_strided((batch, m, n), (c_batch_stride, ldc, 1), c.storage_offset())


def gemm(c, ldc, c_batch_stride, a: View, b: View, batch=1, alpha=1.0, beta=0.0, mul=None) -> None:
    am, bm = a.dense(batch), b.dense(batch)
    prod = torch.einsum("zmk,znk->zmn", am, bm) * alpha
    if mul is not None:
        prod = prod * mul.double()[: prod.shape[1], : prod.shape[2]]
    m, n = prod.shape[1], prod.shape[2]
    if c_batch_stride == 0 and batch > 1:
        out = _strided_out(c, 1, m, n, ldc, 0)
        out.copy_(((out.double() * beta if beta != 0.0 else 0.0) + prod.sum(0, keepdim=True)).to(c.dtype))
        return
    out = _strided_out(c, batch, m, n, ldc, c_batch_stride)
    out.copy_(((out.double() * beta if beta != 0.0 else 0.0) + prod).to(c.dtype))


def rotate_bf16(x, q_t, bias_row=None):
    out = x.double() @ q_t.double()[:, : x.shape[1]].t()
    if bias_row is not None:
        out[:, : bias_row.numel()] += bias_row.double()
    return out.to(torch.bfloat16)


def syrk_accum(cov, x, n_rows, d_in, rows_inner, outer_stride, row_stride, col_stride, mask=None, append_ones=False,
               alpha=1.0, count=None) -> None:
    outer = (n_rows + rows_inner - 1) // rows_inner if n_rows else 0
    rows = x.reshape(-1).as_strided((outer, rows_inner, d_in), (outer_stride, row_stride, col_stride))
    rows = rows.reshape(-1, d_in)[:n_rows].double()
    if append_ones:
        rows = torch.cat([rows, rows.new_ones((n_rows, 1))], dim=1)
    if mask is not None:
        rows = rows * mask.reshape(-1, 1).double()
    cov.add_((alpha * rows.t() @ rows).to(cov.dtype))
    if count is not None:
        count.add_(int(mask.sum().item()) if mask is not None else n_rows)


def im2col(x, conv, append_ones, out_dtype=torch.float32):
    patches = ref.conv_patches(x.double(), conv)  # [b, P, I]
    if append_ones:
        patches = torch.cat([patches, patches.new_ones(patches.shape[:-1] + (1,))], dim=-1)
    return patches.to(out_dtype).contiguous()


def eigh(cov, count, max_sweeps=0, noise_rel=0.0):
    sym = cov.double() / count
    sym = 0.5 * (sym + sym.t())
    evals, evecs = torch.linalg.eigh(sym)
    return evals, evecs.contiguous(), 1


def eigh_small(g, inv_sqrt=False, floor_rel=1e-12):
    sym = 0.5 * (g.double() + g.double().transpose(1, 2))
    evals, evecs = torch.linalg.eigh(sym)
    evals, evecs = evals.flip(-1), evecs.flip(-1)
    if inv_sqrt:
        floor = floor_rel * evals[:, :1].clamp(min=0.0)
        clipped = torch.maximum(evals, floor)
        scale = torch.where(clipped > 0, clipped.rsqrt(), torch.zeros_like(clipped))
        evecs = evecs * scale.unsqueeze(1)
    return evals.float(), evecs.float().contiguous()


def eigh_stats(reset=False):
    return {"factor_first": 0, "fallback": 0, "cholesky_retries": 0}


def rotate_rows_transposed(x, q_t, bias_row=None):
    n, r, d = x.shape
    return rotate_bf16(x.reshape(n * r, d), q_t, bias_row).reshape(n, r, -1).transpose(1, 2).contiguous()


def lambda_rows_accum(lam, gt_t, at_t, scale=1.0) -> None:
    b, o, r = gt_t.shape
    lambda_accum(lam, gt_t.transpose(1, 2).contiguous(), at_t.transpose(1, 2).contiguous(), b, r, scale)


def lambda_accum(lam, gt, at, b, r, scale=1.0) -> None:
    o, ip = lam.shape
    at = at.reshape(b, r, -1)[..., :ip]  # rows may carry zero padding (bf16 engine, odd I')
    g = torch.einsum("bro,bri->boi", gt.reshape(b, r, o).double(), at.double()) * scale
    lam.add_((g * g).sum(0).to(lam.dtype))


def inv_lambda(lam, n_lambda, damping):
    return ref.ekfac_inverse_lambda(lam.double(), torch.tensor([n_lambda], dtype=torch.float64), damping, torch.float32)


def _psg(g, a, append_ones):
    a = a.double()
    if append_ones:
        a = torch.cat([a, a.new_ones(a.shape[:-1] + (1,))], dim=-1)
    return torch.einsum("bro,bri->boi", g.double(), a)


def precondition(g, a, append_ones, q_g, q_a, lam_inv, scale=1.0, out_dtype=torch.float32, q_a_bf16=None, q_g_t_bf16=None,
                 q_a_t_bf16=None):
    psg = _psg(g, a, append_ones)
    out = ref.ekfac_precondition(psg, q_a.double(), q_g.double(), lam_inv.double()) * scale
    if (q_a_bf16 is not None and q_g_t_bf16 is not None and q_a_t_bf16 is not None and out_dtype == torch.bfloat16
            and g.dtype == torch.bfloat16 and g.shape[1] > 1 and g.shape[2] % 8 == 0 and a.shape[2] % 8 == 0
            and a.shape[2] >= 64 and g.shape[2] >= 64):
        out = F.pad(out, (0, q_a_bf16.shape[0] - out.shape[-1]))  # the bf16 engine's padded width (same rule as ops.precondition)
    return out.to(out_dtype).contiguous()


def pairwise_score(scores, col_offset, p, g, a, append_ones, scale=1.0) -> None:
    if hasattr(p, "tiled"):  # TiledQueries: [D/64, Q, 64] -> dense [Q, O, I']
        q, o, ip = p.shape
        p = p.tiled.transpose(0, 1).reshape(q, o, ip)
    psg = _psg(g, a, append_ones)
    block = torch.einsum("qoi,boi->qb", p.double(), psg) * scale
    scores[:, col_offset:col_offset + psg.shape[0]] += block.to(scores.dtype)


def conv2d_cov_geometry(x, conv):
    return None  # no implicit-im2col covariance in the stand-in engine


def conv2d_cov_small(cov, count, x, conv) -> bool:
    return False  # nor the small-patch kernel: the patch path


def conv2d_score_geometry(x_shape, out_channels, conv):
    return None  # the stand-in engine has no implicit-im2col path: the trackers take the patch path


def pairwise_score_conv2d(scores, col_offset, p, g_nchw, x, conv, scale=1.0) -> None:
    """Implicit-im2col entry point: ``p`` is a TiledQueries whose patch axis is ordered (ky, kx, c)."""
    block = ref.conv_pairwise_score(p.dense().double(), x.double(), g_nchw.double(), conv) * scale
    scores[:, col_offset:col_offset + x.shape[0]] += block.to(scores.dtype)


def pairwise_score_rows(scores, col_offset, p, g, a, append_ones, scale=1.0) -> None:
    block = ref.linear_pairwise_score(p.dense().double(), a.double(), g.double(), append_ones) * scale
    scores[:, col_offset:col_offset + g.shape[0]] += block.to(scores.dtype)


def rowwise_dot(out, x, y, weight=None, scale=1.0, accumulate=True) -> None:
    b = x.shape[0]
    prod = x.reshape(b, -1).double() * y.reshape(b, -1).double()
    if weight is not None:
        prod = prod * weight.reshape(1, -1).double()
    value = prod.sum(1) * scale
    if accumulate:
        out.add_(value.to(out.dtype))
    else:
        out.copy_(value.to(out.dtype))


def mul_bcast(x, m, scale=1.0):
    return (x.double() * m.double().reshape((1,) + tuple(x.shape[1:])) * scale).to(torch.float32)


def cast(src, dtype):
    return src.to(dtype).clone()


LEAVES = ("view", "gemm", "rotate_bf16", "rotate_rows_transposed", "lambda_rows_accum", "eigh_stats", "syrk_accum", "im2col", "eigh", "eigh_small", "lambda_accum", "inv_lambda", "precondition",
          "pairwise_score", "conv2d_cov_geometry", "conv2d_cov_small", "conv2d_score_geometry", "pairwise_score_conv2d", "pairwise_score_rows", "rowwise_dot", "mul_bcast", "cast")


class _Setter:
    """Minimal stand-in for pytest's monkeypatch in spawned worker processes (they exit afterwards)."""

    @staticmethod
    def setattr(obj, name, value, raising=True):
        del raising
        setattr(obj, name, value)


def install_in_worker() -> None:
    install(_Setter())


def install(monkeypatch) -> None:
    """Patch the leaf operators and the GPU guards (call from a fixture)."""
    from kronfluence_amd import analyzer, ops

    for name in LEAVES:
        monkeypatch.setattr(ops, name, globals()[name], raising=False)
    monkeypatch.setattr(analyzer.Analyzer, "_require_gpu", staticmethod(lambda cpu: None))
