"""Pairwise-score tracker (reference ``module/tracker/pairwise_score.py``).

During the train pass every tracked layer adds its ``[Q, b]`` contribution into ONE fp32 buffer
``[Q, N_shard]`` resident in HBM (``module.score_sink``), at the column offset of the current batch
(``kf_pairwise_score`` accumulates).  This replaces the reference's per-layer einsum + per-batch
``add_`` over layers + ``.cpu()`` (``score/dot_product.py:105-117``) with one D2H per shard.
Without a sink (direct use of the module API) the tracker falls back to a private ``[Q, b]`` matrix in
``storage["pairwise_score_matrix"]`` as the reference does.
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.base import BaseTracker, QueryBlocks
from kronfluence_amd.utils.constants import (
    ACCUMULATED_PRECONDITIONED_GRADIENT_NAME,
    AGGREGATED_GRADIENT_NAME,
    ACTIVATION_EIGENVECTORS_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    PAIRWISE_SCORE_MATRIX_NAME,
    PRECONDITIONED_GRADIENT_NAME,
)


DIMENSION_NOT_MATCH_ERROR_MSG = (
    "The model does not support token-wise score computation. Set `compute_per_module_scores=True` or "
    "`compute_per_token_scores=False` to avoid this error."
)


class ScoreSink:
    """The fp32 score block of one train shard, resident in HBM and shared by the tracked layers that add into it:
    ``[Q, N_shard]``, or ``[Q, N_shard, T]`` for per-token scores (``T`` is learnt from the first sequence layer)."""

    def __init__(self, num_queries: int, shard_size: int, device: torch.device, per_token: bool = False) -> None:
        self.num_queries, self.shard_size, self.device, self.per_token = num_queries, shard_size, device, per_token
        self.tokens = None if per_token else 1
        self._flat = None if per_token else torch.zeros((num_queries, shard_size), dtype=torch.float32, device=device)

    def matrix(self, tokens: int = 1) -> torch.Tensor:
        """2-D view ``[Q, N_shard * T]`` the kernels accumulate into (sample ``n``, token ``t`` -> column ``n T + t``)."""
        if self.tokens is None:
            self.tokens = tokens
            self._flat = torch.zeros((self.num_queries, self.shard_size * tokens), dtype=torch.float32, device=self.device)
        if tokens != self.tokens:
            raise RuntimeError(DIMENSION_NOT_MATCH_ERROR_MSG)
        return self._flat

    def result(self) -> torch.Tensor:
        if self._flat is None:
            raise RuntimeError("No tracked layer contributed to the scores.")
        if self.per_token and self.tokens != 1:
            return self._flat.view(self.num_queries, self.shard_size, self.tokens)
        return self._flat


# Low-rank query gradients ``[L_q [O, k], R_q [k, I']]`` against a train batch: the reference hands "qik,qko,b...i,b...o->qb" to
# opt_einsum (module/linear.py:83-99), which picks the cheaper of two exact orders per call.  So does this tracker, per hook call,
# with bytes in the model as well as flops (``_low_rank_plan``):
#   * EXPAND  P_q = L_q R_q, then the dense score path: 2 O I' (k + b) flops per query and batch plus one write and one read of
#     the O I' block -- right for narrow layers and large train batches (BERT: O I' = 0.6 M, b = 512).  If all held queries of a
#     layer fit ``LOW_RANK_CACHE_BYTES`` the expansion is done once per train pass and kept; otherwise it is re-done per train
#     batch in blocks of at most ``low_rank_block_bytes()`` (storage is what low rank buys).
#   * FACTORED  scores[q, n] = sum_{r, k} (G_n L_q)[r, k] (A'_n R_q^T)[r, k]: 2 R k (O + I') flops per pair and NO O I' block --
#     right when the block dominates: one row per sample (always), or wide layers against small batches (a Llama-3-8B MLP
#     projection, O I' = 58.7 M, k = 64, 16 sequences: expanding costs 0.23 GB per query and batch, 3x the time of the factored
#     flops).  Two tall NT GEMMs on the LDS-DMA engine + ``kf_lowrank_rows_dot``.
LOW_RANK_CACHE_BYTES = 1 << 30  # per layer: only small expansions are kept for the whole train pass
LOW_RANK_FACTORED_BYTES = 2 << 30  # bf16 bytes of ONE of the two [b R, q k] products per query chunk


def low_rank_block_bytes(device) -> int:
    """fp32 bytes of one expanded query block: a third of the HBM that is free right now, at most 32 GiB (the bf16 copy for the
    score GEMM is another half of it)."""
    free = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else 8 << 30
    return int(max(256 << 20, min(32 << 30, free // 3)))


def unpadded_queries(module, preconditioned):
    """The held query gradients at the reference's width: strips the zero columns the bf16 preconditioner appends for an
    odd ``I'`` (``module.query_padding``).  For every reader that is not one of the padded score kernels."""
    pad = getattr(module, "query_padding", 0)
    if pad and isinstance(preconditioned, QueryBlocks):
        return QueryBlocks([b[..., :b.shape[-1] - pad].contiguous() for b in preconditioned.blocks])
    if pad and torch.is_tensor(preconditioned):
        return preconditioned[..., :preconditioned.shape[-1] - pad].contiguous()
    return preconditioned


def dense_queries(preconditioned, score_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """The held query gradients as one dense ``[Q, O, I']`` tensor (expands low-rank factors)."""
    if isinstance(preconditioned, (TiledQueries, QueryBlocks)):
        preconditioned = preconditioned.dense()
    if isinstance(preconditioned, list):
        dense = ops.low_rank_product(preconditioned[0], preconditioned[1])
        return ops.cast(dense, torch.bfloat16) if score_dtype == torch.bfloat16 else dense
    if preconditioned.dtype not in (torch.float32, torch.bfloat16):
        return preconditioned.to(torch.float32)
    return preconditioned


class TiledQueries:
    """The held bf16 query gradients of one layer in the layout the bf16 score contraction streams best: k-tile-major
    ``[D'/64, Q, 64]`` with ``D' = O * I'_pad`` (``I'`` zero-padded to a multiple of 8, see ``PAD_PATCH_AXIS``).  Built
    ONCE per train pass, in query chunks, and it REPLACES the dense ``[Q, O, I']`` block in ``module.storage``: at BERT /
    GPT-2 scale the dense block is 150-175 GB, a second (let alone a padded third) copy does not fit 288 GB."""

    CHUNK = 64  # queries converted at a time (bounds the padded / permuted temporary)

    def __init__(self, dense: torch.Tensor, pad: int, conv_channels: int = 0, prepadded: int = 0) -> None:
        """``pad``: zero columns to append; ``prepadded``: trailing zero columns ``dense`` already carries (the bf16
        preconditioner's output for an odd ``I'``).  ``conv_channels = C > 0``: the patch axis ``(c, ky, kx)`` of a
        convolution's gradient is re-ordered to ``(ky, kx, c)``, the order in which the implicit-im2col kernel produces
        per-sample gradients (``kf_pairwise_score_conv2d``)."""
        blocks = dense.blocks if isinstance(dense, QueryBlocks) else [dense]
        q, o, ip = (sum(b.shape[0] for b in blocks),) + tuple(blocks[0].shape[1:])
        self.num_queries, self.rows, self.conv_channels = q, o, conv_channels
        self.pad = pad + prepadded  # total zero columns relative to the reference's I'
        assert dense.dtype == torch.bfloat16 and not (self.pad and conv_channels)
        taps = ip // conv_channels if conv_channels else 0
        self.conv_padded = conv_channels + (-conv_channels) % 8  # channels incl. the zero channels the kernel adds
        self.width = taps * self.conv_padded if conv_channels else ip + pad
        d = o * self.width
        assert d % 64 == 0
        self.tiled = torch.empty((d // 64, q, 64), dtype=torch.bfloat16, device=dense.device)
        row = 0
        for source in blocks:
            for start in range(0, source.shape[0], self.CHUNK):
                block = source[start:start + self.CHUNK]
                if pad:
                    block = torch.nn.functional.pad(block, (0, pad))
                if conv_channels:
                    block = block.reshape(block.shape[0], o, conv_channels, taps).transpose(2, 3)
                    block = torch.nn.functional.pad(block, (0, self.conv_padded - conv_channels))
                self.tiled[:, row:row + block.shape[0]] = block.reshape(block.shape[0], d // 64, 64).transpose(0, 1)
                row += block.shape[0]

    @property
    def shape(self):
        return (self.num_queries, self.rows, self.width)

    def dense(self) -> torch.Tensor:
        """Back to the reference's ``[Q, O, I']`` (rare paths only)."""
        full = self.tiled.transpose(0, 1).reshape(self.num_queries, self.rows, self.width)
        if self.conv_channels:
            c, cp = self.conv_channels, self.conv_padded
            full = full.reshape(self.num_queries, self.rows, self.width // cp, cp)[..., :c].transpose(2, 3)
            return full.reshape(self.num_queries, self.rows, -1).contiguous()
        return full[:, :, :self.width - self.pad].contiguous()


class PairwiseScoreTracker(BaseTracker):
    _expanded = None  # (left factor, dense tensor): per-pass cache of expanded low-rank queries

    LOW_RANK_ROW_CHUNK = 1 << 28   # fp32 elements of one [q_c, b, k] product of the one-row factored contraction

    def _score_low_rank_rows(self, left: torch.Tensor, right: torch.Tensor, g: torch.Tensor, a: torch.Tensor, ones: bool,
                             scores: torch.Tensor, offset: int, scale: float) -> None:
        """One row per sample: ``scores[q, n] += sum_k (g_n . L_q[:, k]) (R_q[k, :] . a'_n)`` -- the reference's
        ``"qik,qko,bi,bo->qb"`` (module/linear.py:83-99) contracted WITHOUT expanding ``P_q = L_q R_q``: two skinny batched
        GEMMs and a k-long row dot, ``2 k (O + I')`` flops per pair instead of ``2 O I'``."""
        q, o, k = left.shape
        ip = right.shape[2]
        b = g.shape[0]
        gq, aq = g.reshape(b, o).contiguous(), a.reshape(b, -1).contiguous()   # any float dtype: the GEMM loaders convert
        i = aq.shape[1]                                                         # (bias: the loader supplies the ones column)
        if self._low_rank_f32 is None or self._low_rank_f32[0] is not left:
            self._low_rank_f32 = (left, left.float().contiguous(), right.float().contiguous())
        _, lf, rf = self._low_rank_f32
        dev = g.device
        step = max(1, self.LOW_RANK_ROW_CHUNK // max(1, b * k))   # queries per chunk: bounds the two [q_c, b, k] temporaries
        for first in range(0, q, step):
            lc, rc = lf[first:first + step], rf[first:first + step]
            qc = lc.shape[0]
            u = ops._bmm((qc, b, k), ops.view(gq, 0, o, 1, b, o), ops.view(lc, o * k, 1, k, k, o), qc, dev)          # g L_q
            v = ops._bmm((qc, b, k), ops.view(aq, 0, i, 1, b, i, ones_k=ones), ops.view(rc, k * ip, ip, 1, k, ip), qc, dev)   # a' R_q^T
            block = torch.empty(qc * b, dtype=torch.float32, device=dev)
            ops.rowwise_dot(block, u.reshape(qc * b, k), v.reshape(qc * b, k), scale=scale, accumulate=False)
            scores[first:first + qc, offset:offset + b].add_(block.view(qc, b))

    _low_rank_f32 = None  # (left as held, left fp32, right fp32): per-pass cache for the factored contraction
    _low_rank_stacked = None  # (left as held, L^T stacked [Q k, O], R stacked [Q k, W], bias column [Q k]): sequence layers

    # effective rates of the plan's cost model (measured orders of magnitude, MI355X): tall bf16 NT GEMMs, HBM streams
    _PLAN_FLOPS, _PLAN_BYTES = 8.0e14, 3.0e12

    def _low_rank_plan(self, left: torch.Tensor, right: torch.Tensor, g: torch.Tensor, a: torch.Tensor, ones: bool) -> str:
        """``"factored"`` or ``"expand"`` for this hook call (see the module comment): estimated seconds of both exact orders."""
        q, o, k = left.shape
        ip = right.shape[2]
        b, r = g.shape[0], g.shape[1]
        if r == 1:
            return "factored"
        i = a.shape[-1]
        # the sequence form runs on the bf16 engines and holds U, V in bf16: only when every operand already IS bf16 (the low-precision
        # presets) -- with fp32 factors / gradients the expanded order keeps the reference's fp32 arithmetic, whatever the cost model
        # says
        eligible = (g.is_cuda and o % 8 == 0 and i % 8 == 0 and k % 8 == 0 and o >= 64 and i >= 64 and ip == i + int(ones)
                    and b <= 65535 and left.dtype == right.dtype == g.dtype == a.dtype == torch.bfloat16)
        if not eligible:
            return "expand"
        block = float(q) * o * ip
        cached = block * 4 <= LOW_RANK_CACHE_BYTES
        factored = 2.0 * b * r * q * k * (o + ip) / self._PLAN_FLOPS + 6.0 * b * r * q * k * 2 / self._PLAN_BYTES
        expand = 2.0 * block * b / self._PLAN_FLOPS + block * 2 / self._PLAN_BYTES
        if not cached:   # re-expanded on every train batch: fp32 product written, read, bf16 copy written
            expand += 2.0 * block * k / self._PLAN_FLOPS + block * 10 / self._PLAN_BYTES
        return "factored" if factored < expand else "expand"

    def _score_low_rank_sequences(self, left: torch.Tensor, right: torch.Tensor, g: torch.Tensor, a: torch.Tensor, ones: bool,
                                  scores: torch.Tensor, offset: int, scale: float) -> None:
        """Several rows per sample, factored: ``U = G [L_1 .. L_Q]`` and ``V = A' [R_1^T .. R_Q^T]`` as two tall NT GEMMs over
        all rows of the batch (bias column of ``R`` in the epilogue), then ``kf_lowrank_rows_dot`` over (row, k) per pair --
        ``"qik,qko,b...i,b...o->qb"`` of module/linear.py:83-99 without any ``[O, I']`` block."""
        q, o, k = left.shape
        ip = right.shape[2]
        b, r = g.shape[0], g.shape[1]
        i = a.shape[-1]
        if self._low_rank_stacked is None or self._low_rank_stacked[0] is not left:
            lt = left.transpose(1, 2).to(torch.bfloat16).reshape(q * k, o).contiguous()
            width = i + (-i) % 8   # the GEMM reads the first I columns; the bias column (if any) travels as a row addend
            rt = right[..., :i].to(torch.bfloat16).reshape(q * k, i)
            rt = torch.nn.functional.pad(rt, (0, width - i)).contiguous() if width != i else rt.contiguous()
            bias = right[..., i].to(torch.float32).reshape(q * k).contiguous() if ones else None
            self._low_rank_stacked = (left, lt, rt, bias)
        _, lt, rt, bias = self._low_rank_stacked
        g2 = (g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)).reshape(b * r, o)
        a2 = (a if a.dtype == torch.bfloat16 else a.to(torch.bfloat16)).reshape(b * r, i)
        step = max(1, LOW_RANK_FACTORED_BYTES // max(1, b * r * k * 2))
        for first in range(0, q, step):
            rows = slice(first * k, min(q, first + step) * k)
            qc = (rows.stop - rows.start) // k
            u = ops.rotate_bf16(g2, lt[rows])
            v = ops.rotate_bf16(a2, rt[rows], bias[rows] if bias is not None else None)
            ops.lowrank_rows_dot(scores[first:first + qc], offset, u, v, b, r, qc, k, scale=scale)

    def _query_blocks(self, preconditioned):
        """Yields ``(first_row, dense [q_c, O, I'])`` covering the held queries."""
        if isinstance(preconditioned, QueryBlocks):
            first = 0
            for block in preconditioned.blocks:
                yield first, dense_queries(block)
                first += block.shape[0]
            return
        if not isinstance(preconditioned, list):
            yield 0, dense_queries(preconditioned)
            return
        left, right = preconditioned
        score_dtype = left.dtype if left.dtype == torch.bfloat16 else torch.float32
        q, o, ip = left.shape[0], left.shape[1], right.shape[2]
        per_query = o * ip * 4
        if q * per_query <= LOW_RANK_CACHE_BYTES:
            if self._expanded is None or self._expanded[0] is not left:
                self._expanded = (left, dense_queries(preconditioned, score_dtype))
            yield 0, self._expanded[1]
            return
        step = max(1, low_rank_block_bytes(left.device) // per_query)
        for start in range(0, q, step):
            yield start, dense_queries([left[start:start + step], right[start:start + step]], score_dtype)

    # bf16 layers whose augmented input axis I' is not a multiple of 8 -- a first conv layer (3*3*3 = 27 patch columns),
    # or ANY Linear with a bias applied to sequences (I' = I + 1: BERT / GPT-2 shapes) -- would fall back to the fp32
    # engine for the per-sample gradients and the score contraction (about 8x slower than the bf16 MFMA engine).
    # Instead that axis is zero-padded to the next multiple of 8 on both sides of the contraction (the bias column of
    # ones is materialised first): <P_q, g_n> is unchanged, P is padded once per train pass (inside ``TiledQueries``).
    PAD_PATCH_AXIS = True

    def _fast_layout(self, preconditioned, g: torch.Tensor, a: torch.Tensor, ones: bool):
        """-> ``(TiledQueries, a, ones)`` when the bf16 k-tile-major score engine applies to this layer (bf16 queries and
        factors, several rows per sample, ``O % 8 == 0``, ``O * I'_pad % 64 == 0``), else ``None``.  The first call of a
        train pass converts the held dense block and replaces it in ``module.storage``."""
        tiled = preconditioned if isinstance(preconditioned, TiledQueries) else None
        if tiled is not None and tiled.conv_channels:  # laid out for the implicit-im2col kernel: back to the reference's
            preconditioned, tiled = tiled.dense(), None
        if tiled is None:
            if (not (torch.is_tensor(preconditioned) or isinstance(preconditioned, QueryBlocks))
                    or preconditioned.dtype != torch.bfloat16 or g.shape[1] == 1
                    or g.dtype != torch.bfloat16 or a.dtype != torch.bfloat16 or g.shape[-1] % 8 != 0):
                return None
            width = a.shape[-1] + int(ones)
            prepadded = self.module.query_padding
            pad = ((-width) % 8 if self.PAD_PATCH_AXIS else 0) - prepadded
            if (pad < 0 or preconditioned.shape[-1] != width + prepadded or (width + prepadded + pad) % 8 != 0
                    or (g.shape[-1] * (width + prepadded + pad)) % 64 != 0):
                return None
            tiled = TiledQueries(preconditioned, pad, prepadded=prepadded)
            self.module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = tiled  # the dense block is released
        if tiled.pad or ones:  # materialise the ones column and the zero padding on the train side as well
            parts = [a]
            if ones:
                parts.append(a.new_ones(a.shape[:-1] + (1,)))
            if tiled.pad:
                parts.append(a.new_zeros(a.shape[:-1] + (tiled.pad,)))
            a = torch.cat(parts, dim=-1)
        return tiled, a, False

    # Second-generation bf16 score path (csrc/kf_score_v2.hip): LDS-DMA fed kernels, implicit im2col for convolutions
    # (no patch tensor, no transposed gradient copy), in-kernel bias column / padding for Linear layers on sequences.
    SCORE_V2 = True

    def _score_v2(self, preconditioned, activation: torch.Tensor, output_gradient: torch.Tensor, scores: torch.Tensor,
                  offset: int) -> bool:
        """Scores this layer on the v2 kernels when its shapes and dtypes allow; ``False`` -> caller takes the v1 path."""
        module, original = self.module, self.module.original_module
        tiled = preconditioned if isinstance(preconditioned, TiledQueries) else None
        if not self.SCORE_V2 or output_gradient.dtype != torch.bfloat16:
            return False
        if tiled is None and not ((torch.is_tensor(preconditioned) or isinstance(preconditioned, QueryBlocks))
                                  and preconditioned.dtype == torch.bfloat16):
            return False
        if isinstance(original, nn.Conv2d) and activation.dim() == 4 and output_gradient.dim() == 4:
            geometry = ops.conv2d_score_geometry(activation.shape, output_gradient.shape[1], original)
            if geometry is None:
                return False
            c, o, (k1, k2) = activation.shape[1], output_gradient.shape[1], original.kernel_size
            ip = c * k1 * k2
            if tiled is None:
                if tuple(preconditioned.shape[1:]) != (o, ip):
                    return False
                tiled = TiledQueries(preconditioned, 0, conv_channels=c)
                module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = tiled  # the dense block is released
            elif tiled.conv_channels != c or tiled.shape[1:] != (o, (c + (-c) % 8) * k1 * k2):
                return False
            x = activation if activation.dtype == torch.bfloat16 else activation.to(torch.bfloat16)
            ops.pairwise_score_conv2d(scores, offset, tiled, output_gradient, x, original, scale=module.gradient_scale)
            return True
        if isinstance(original, nn.Linear) and activation.dim() >= 3:
            o, i = output_gradient.shape[-1], activation.shape[-1]
            rows = activation.numel() // (activation.shape[0] * i)
            width = i + int(original.bias is not None)
            padded = width + (-width) % 8
            if rows % 64 != 0 or o % 8 != 0 or i % 8 != 0 or (o * padded) % 64 != 0 or activation.shape[0] > 65535:
                return False
            if tiled is None:
                prepadded = module.query_padding
                if tuple(preconditioned.shape[1:]) != (o, width + prepadded) or prepadded not in (0, padded - width):
                    return False
                tiled = TiledQueries(preconditioned, padded - width - prepadded, prepadded=prepadded)
                module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = tiled
            elif tiled.conv_channels or tiled.shape[1:] != (o, padded):
                return False
            a = activation if activation.dtype == torch.bfloat16 else activation.to(torch.bfloat16)
            b = activation.shape[0]
            self._score_rows_paired(scores, offset, tiled, output_gradient.reshape(b, rows, o), a.reshape(b, rows, i),
                                    original.bias is not None, module.gradient_scale)
            return True
        return False

    # Train micro-batches of a sequence layer are scored IN PAIRS when they are small: the score GEMM reads the whole P of the
    # layer (Q O I' x 2 bytes: 1.2 - 4.8 GB per GPT-2 layer at 1 024 queries) once per launch, so at b = 128 sequences its
    # arithmetic intensity is 128 flop/byte -- HBM bound (measured 5.5 TB/s on P alone, profiles/README.md round 4), not MFMA
    # bound.  The hooked (gradient, activation) pair of one batch is therefore HELD (by reference; version-checked) until the
    # next batch's hook of the same layer arrives, and both go through ONE call (kf_pairwise_score_rows2: b = 256, half the P
    # traffic per pair, 256 x 256 tiles).  An odd last batch, a layer used twice per pass, non-adjacent score columns: the held
    # batch is scored alone.  Bounded by ``PAIR_HOLD_FRACTION`` of the device memory over all layers.
    PAIR_MAX_BATCH = 192          # micro-batches above this size fill the 256-row tile well enough on their own
    PAIR_MIN_QUERIES = 256
    PAIR_HOLD_FRACTION = 0.10
    _pair_held = None             # (scores, offset, g, a, versions, ones, scale, bytes)
    _pair_bytes_all_layers = [0]  # shared by every tracker of the process

    def _score_rows_paired(self, scores, offset, tiled, g, a, ones, scale) -> None:
        held = self._pair_held
        b = g.shape[0]
        if held is not None:
            h_scores, h_offset, h_g, h_a, versions, h_ones, h_scale, _ = held
            adjacent = (h_scores is scores and h_offset + h_g.shape[0] == offset and h_ones == ones and h_scale == scale
                        and tuple(h_g.shape[1:]) == tuple(g.shape[1:]) and tuple(h_a.shape[1:]) == tuple(a.shape[1:])
                        and h_g.shape[0] + b <= 65535)
            if adjacent:
                self._check_held_versions(h_g, h_a, versions)
                self._drop_held()
                ops.pairwise_score_rows(scores, h_offset, tiled, h_g, h_a, ones, scale=scale, second=(g, a))
                return
            self._flush_pair(tiled)
        nbytes = (g.numel() + a.numel()) * 2
        budget = self._pair_budget(g.device)
        if (self.module.score_sink is not None and b <= self.PAIR_MAX_BATCH and tiled.shape[0] >= self.PAIR_MIN_QUERIES
                and self._pair_bytes_all_layers[0] + nbytes <= budget):
            self._pair_held = (scores, offset, g, a, (g._version, a._version), ones, scale, nbytes)
            self._pair_bytes_all_layers[0] += nbytes
            return
        ops.pairwise_score_rows(scores, offset, tiled, g, a, ones, scale=scale)

    def _pair_budget(self, device: torch.device) -> float:
        """Bytes of hooked tensors all layers together may hold across a batch boundary."""
        if device.type != "cuda":
            return 0.0
        return self.PAIR_HOLD_FRACTION * torch.cuda.get_device_properties(device).total_memory

    def _check_held_versions(self, g, a, versions) -> None:
        if (g._version, a._version) != versions:
            raise RuntimeError(
                f"A tensor hooked at module '{self.module.name}' was modified in place while the score tracker held it for the next "
                "train batch; make the offending operation out-of-place (or set PairwiseScoreTracker.PAIR_MAX_BATCH = 0).")

    def _drop_held(self) -> None:
        if self._pair_held is not None:
            self._pair_bytes_all_layers[0] -= self._pair_held[-1]
            self._pair_held = None

    def _flush_pair(self, tiled=None) -> None:
        """Scores a held micro-batch on its own (end of the train pass, or the next batch could not be paired with it)."""
        held = self._pair_held
        if held is None:
            return
        scores, offset, g, a, versions, ones, scale, _ = held
        self._check_held_versions(g, a, versions)
        self._drop_held()
        if tiled is None:
            tiled = self.module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
        if not isinstance(tiled, TiledQueries):   # the queries this batch was hooked against are gone: never drop scores silently
            raise RuntimeError(f"Module '{self.module.name}' holds a train micro-batch to score but no tiled query gradients.")
        ops.pairwise_score_rows(scores, offset, tiled, g, a, ones, scale=scale)

    def register_hooks(self) -> None:
        module = self.module
        storage = module.storage

        @torch.no_grad()
        def forward_hook(mod: nn.Module, inputs: Tuple[torch.Tensor], outputs: torch.Tensor) -> None:
            del mod
            self._cache_activation(inputs[0].detach())
            self.cached_hooks.append(outputs.register_hook(backward_hook))

        @torch.no_grad()
        def backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            if module.score_sink is None:   # direct use of the module API: the caller reads storage right after backward()
                score_batch(activation, output_gradient)
                return
            # the layer's score kernels run beside the rest of the model's backward pass when memory allows (BaseTracker._run_beside);
            # ``finalize_all_iterations`` joins the side stream before anyone reads the score block
            self._run_beside(output_gradient.device, (activation, output_gradient), lambda: score_batch(activation, output_gradient))

        def score_batch(activation: torch.Tensor, output_gradient: torch.Tensor) -> None:
            preconditioned = storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
            if preconditioned is None:
                raise RuntimeError(f"Module '{module.name}' holds no preconditioned query gradient.")
            num_queries = preconditioned[0].shape[0] if isinstance(preconditioned, list) else preconditioned.shape[0]
            if isinstance(preconditioned, TiledQueries) and module.per_sample_gradient_process_fnc is not None:
                preconditioned = preconditioned.dense()
            batch = output_gradient.shape[0]
            per_token = module.score_args.compute_per_token_scores and activation.dim() == 3 and module.score_sink is not None
            if module.score_sink is not None:
                sink, offset = module.score_sink
                tokens = activation.shape[1] if per_token else 1
                scores, offset = sink.matrix(tokens), offset * tokens
            else:
                scores = torch.zeros((num_queries, batch), dtype=torch.float32, device=output_gradient.device)
                offset = 0
                accumulate_into = storage[PAIRWISE_SCORE_MATRIX_NAME] if module.factor_args.has_shared_parameters else None
                if accumulate_into is not None and accumulate_into.shape == scores.shape:
                    scores = accumulate_into
                storage[PAIRWISE_SCORE_MATRIX_NAME] = scores
            if (module.per_sample_gradient_process_fnc is None and not per_token and not module.queries_in_eigenbasis
                    and self._score_v2(preconditioned, activation, output_gradient.detach(), scores, offset)):
                return
            if module.per_sample_gradient_process_fnc is None:
                g, a, ones = module.gradient_factors(activation, output_gradient.detach())
                if per_token:  # "qio,bti,bto->qbt" (linear.py:100-111): every token is a rank-one "sample"
                    g, a = g.reshape(-1, 1, g.shape[-1]), a.reshape(-1, 1, a.shape[-1])
                if module.queries_in_eigenbasis:  # see PreconditionTracker.EIGENBASIS_QUERIES
                    # rotated row by row, keeping the R axis: the held queries had one row per sample, the train
                    # batch may have several (single-token queries against sequence batches)
                    n, r = g.shape[0], g.shape[1]
                    g = ops.matmul_nn(g.reshape(n * r, -1), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME)).reshape(n, r, -1)
                    a = ops.matmul_nn(a.reshape(n * r, -1), self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME),
                                      append_ones=ones).reshape(n, r, -1)
                    ones = False
                if isinstance(preconditioned, list) and self._low_rank_plan(preconditioned[0], preconditioned[1], g, a, ones) == "factored":
                    score = self._score_low_rank_rows if g.shape[1] == 1 else self._score_low_rank_sequences
                    score(preconditioned[0], preconditioned[1], g, a, ones, scores, offset, module.gradient_scale)
                    return
                fast = self._fast_layout(preconditioned, g, a, ones)
                if fast is not None:
                    tiled, a_in, ones_in = fast
                    ops.pairwise_score(scores, offset, tiled, g, a_in, ones_in, scale=module.gradient_scale)
                else:
                    if module.query_padding and not isinstance(preconditioned, (TiledQueries, list)):
                        # the padded-width kernels do not apply to this batch: strip the zero columns ONCE and keep the
                        # stripped blocks (a per-batch ``unpadded_queries`` copied the whole query set on every hook call)
                        preconditioned = unpadded_queries(module, preconditioned)
                        storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = preconditioned
                        module.query_padding = 0
                    for first, block in self._query_blocks(unpadded_queries(module, preconditioned)):
                        rows = scores[first:first + block.shape[0]]
                        ops.pairwise_score(rows, offset, block, g, a, ones, scale=module.gradient_scale)
            else:
                # post-processed gradient (pairwise_score.py:41-45): contract the materialised gradient
                psg = module.compute_per_sample_gradient(activation, output_gradient.detach()).contiguous()
                b, o, ip = psg.shape
                for first, block in self._query_blocks(unpadded_queries(module, preconditioned)):
                    block = block if block.dtype == torch.float32 else ops.cast(block, torch.float32)
                    q = block.shape[0]
                    ops.gemm(scores[first:first + q, offset:offset + b], scores.shape[1], 0,
                             ops.view(block.contiguous(), 0, o * ip, 1, q, o * ip),
                             ops.view(psg, 0, o * ip, 1, b, o * ip), alpha=module.gradient_scale, beta=1.0)

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    def finalize_iteration(self) -> None:
        self.clear_all_cache()

    def exist(self) -> bool:
        return self.module.storage[PAIRWISE_SCORE_MATRIX_NAME] is not None or self.module.score_sink is not None

    def accumulate_iterations(self) -> None:
        self.release_memory()

    @torch.no_grad()
    def finalize_all_iterations(self) -> None:
        """With ``aggregate_train_gradients`` the ``GradientTracker`` left the summed train gradient in storage:
        its dot product with every held query gradient is column 0 of the sink (reference ``pairwise_score.py:119-133``)."""
        module, storage = self.module, self.module.storage
        side = self._SIDE.get("stream") if self._side_done is not None else None
        if side is not None:
            with torch.cuda.stream(side):
                self._flush_pair()   # a train micro-batch still waiting for a partner (odd number of batches)
        else:
            self._flush_pair()
        self._join_side()            # the score block is complete for whoever reads it next
        summed = storage[AGGREGATED_GRADIENT_NAME]
        if summed is not None and module.score_sink is not None:
            preconditioned = storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
            if preconditioned is None:
                raise RuntimeError(f"Module '{module.name}' holds no preconditioned query gradient.")
            preconditioned = dense_queries(unpadded_queries(module, preconditioned)).contiguous()
            if preconditioned.dtype != torch.float32:
                preconditioned = ops.cast(preconditioned, torch.float32)
            summed = summed.to(torch.float32).contiguous()
            _, o, ip = summed.shape
            if module.queries_in_eigenbasis:  # queries are held as M_q: rotate the summed gradient instead
                q_a, q_g = self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME)
                t1 = torch.empty((o, ip), dtype=torch.float32, device=summed.device)
                ops.gemm(t1, ip, 0, ops.view(summed, 0, ip, 1, o, ip), ops.view(q_a, 0, 1, ip, ip, ip))
                rotated = torch.empty((1, o, ip), dtype=torch.float32, device=summed.device)
                ops.gemm(rotated, ip, 0, ops.view(q_g, 0, 1, o, o, o), ops.view(t1, 0, 1, ip, ip, o))
                summed = rotated
            sink, offset = module.score_sink
            scores = sink.matrix(1)
            q = preconditioned.shape[0]
            ops.gemm(scores[:, offset:offset + 1], scores.shape[1], 0, ops.view(preconditioned, 0, o * ip, 1, q, o * ip),
                     ops.view(summed, 0, o * ip, 1, 1, o * ip), beta=1.0)
            storage[AGGREGATED_GRADIENT_NAME] = None
        self.module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = None
        self.module.storage[PRECONDITIONED_GRADIENT_NAME] = None
        self.module.score_sink = None
        self._expanded = None
        self._low_rank_f32 = None
        self._low_rank_stacked = None
        self.clear_all_cache()

    def release_memory(self) -> None:
        self._drop_held()
        self.clear_all_cache()
        self.module.storage[PAIRWISE_SCORE_MATRIX_NAME] = None
