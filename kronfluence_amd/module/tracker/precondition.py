"""Preconditioned query gradients (reference ``module/tracker/precondition.py``).

The hook hands the query batch's activation / output-gradient factors to ``kf_precondition``
(per-sample gradient + EK-FAC preconditioner + scale in one call chain on the MFMA engine); results
stay resident in HBM.  With ``query_gradient_low_rank`` only rank-k factors are kept.
"""

from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.factor.config import FactorConfig
from kronfluence_amd.module.tracker.base import BaseTracker, QueryBlocks, QueryBuffer
from kronfluence_amd.utils.comm import exchange
from kronfluence_amd.utils.state import force_exchanges
from kronfluence_amd.utils.constants import (
    ACCUMULATED_PRECONDITIONED_GRADIENT_NAME,
    AGGREGATED_GRADIENT_NAME,
    ACTIVATION_EIGENVECTORS_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    LAMBDA_MATRIX_NAME,
    PRECONDITIONED_GRADIENT_NAME,
)


class PreconditionTracker(BaseTracker):
    # For 2-D activations (one row per sample) the EK-FAC preconditioned query gradient is kept IN THE
    # EIGENBASIS: M_q = (Qg^T g_q Qa) o Lambda^-1 instead of P_q = Qg M_q Qa^T.  The train pass then rotates
    # its (rank-one) gradient factors instead -- <P_q, g_n> = <M_q, (G_n Qg) (x) (A'_n Qa)> exactly -- which
    # costs 2 b (O^2 + I'^2) flops per batch instead of 2 Q O I' (O + I') per query chunk (the back-rotation
    # was 70 % of the MNIST-MLP pairwise stage).  `storage["preconditioned_gradient"]` then holds M_q; set this
    # to False to store the reference's P_q.
    EIGENBASIS_QUERIES = True

    # Multi-GPU: the all-gather of a layer's block (C4, reference precondition.py:181-201) is ISSUED from the backward hook
    # that produced it, asynchronously -- it travels over xGMI while autograd runs the remaining layers' backward and their
    # preconditioners -- and ``synchronize`` only waits for it and interleaves.  All ranks run the same graph, so the
    # collectives are issued in the same order everywhere.  Opt-in per stage: only the pairwise query loop, which follows every
    # query batch with ``synchronize`` and whose strided sampler gives all ranks equal blocks, sets
    # ``module.async_query_gather`` (multi-GPU self-influence with measurement runs in this mode too and exchanges nothing).
    ASYNC_QUERY_GATHER = True
    _pending = None  # (work, gathered, local) of the all-gather in flight
    _held_layout = None  # (block shape, query_padding, queries_in_eigenbasis) of the blocks accumulated so far

    def _out_dtype(self) -> torch.dtype:
        """``score_dtype`` of the reference (precondition.py:73): bf16 keeps P in bf16 for the bf16 MFMA
        score contraction; everything else is held in fp32."""
        return torch.bfloat16 if self.module.score_args.score_dtype == torch.bfloat16 else torch.float32

    _bf16_q = None

    def _bf16_eigenvectors(self):
        """bf16 operands ``(Q_A, Q_G^T, Q_A^T, Q_G, bias row)`` for ``precondition_dtype == bf16`` (the reference casts the eigenvectors
        to that dtype in ``Ekfac.prepare``, factor/config.py:323-328): ``Q_A`` and ``Q_A^T`` zero-padded to ``[W, W]`` with ``W = I'``
        rounded up to a multiple of 8 (see ``ops.precondition``), ``Q_G`` as stored when it is stored in bf16, the bias row
        ``Q_A[I]`` in fp32 for layers with a bias; five ``None`` otherwise.  Stored matrices that already have the layout asked for
        (bf16, ``I' % 8 == 0``) are used as they are -- no copy."""
        args = self.module.score_args
        if args.precondition_dtype != torch.bfloat16 or args.score_dtype != torch.bfloat16:
            return None, None, None, None, None
        storage = self.module.storage
        source, source_g = storage[ACTIVATION_EIGENVECTORS_NAME], storage[GRADIENT_EIGENVECTORS_NAME]
        if self._bf16_q is None or self._bf16_q[0] is not source:
            pad = (-source.shape[0]) % 8
            if pad == 0 and source.dtype == torch.bfloat16 and source.is_contiguous():
                q_a = source
            else:
                q_a = torch.nn.functional.pad(source, (0, pad, 0, pad)).to(torch.bfloat16).contiguous()
            q_a_t = torch.nn.functional.pad(source.t(), (0, pad, 0, pad)).to(torch.bfloat16).contiguous()
            q_g = source_g if (source_g.dtype == torch.bfloat16 and source_g.is_contiguous()) else source_g.to(torch.bfloat16).contiguous()
            bias_row = source[-1].to(torch.float32).contiguous() if self.module.has_bias else None
            self._bf16_q = (source, q_a, source_g.t().contiguous().to(torch.bfloat16), q_a_t, q_g, bias_row)
        return self._bf16_q[1:]

    def _store(self, preconditioned: torch.Tensor, from_hook: bool = False) -> None:
        """Keeps the ``[q, O, I']`` block, or -- with ``query_gradient_low_rank = k < min(O, I')`` -- its rank-k factors
        ``[left [q,O,k], right [q,k,I']]`` (reference ``precondition.py:19-75``; ``use_full_svd`` buys two more
        subspace iterations instead of a dense SVD)."""
        args = self.module.score_args
        rank = args.query_gradient_low_rank
        if rank is not None and self.module.query_padding:
            preconditioned = preconditioned[..., :preconditioned.shape[-1] - self.module.query_padding].contiguous()
            self.module.query_padding = 0
        if rank is not None and min(preconditioned.shape[1:]) > rank:
            dense = preconditioned if preconditioned.dtype == torch.float32 else ops.cast(preconditioned, torch.float32)
            left, right = ops.low_rank_factors(dense, rank, power_iterations=4 if args.use_full_svd else 2)
            if self._out_dtype() != torch.float32:
                left, right = ops.cast(left, self._out_dtype()), ops.cast(right, self._out_dtype())
            self.module.storage[PRECONDITIONED_GRADIENT_NAME] = [left, right]
            return
        if preconditioned.dtype != self._out_dtype():
            preconditioned = preconditioned.to(self._out_dtype())
        preconditioned = preconditioned.contiguous()
        self.module.storage[PRECONDITIONED_GRADIENT_NAME] = preconditioned
        if (from_hook and self.ASYNC_QUERY_GATHER and self.module.async_query_gather and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size() > 1 or force_exchanges())):
            if self._pending is not None:   # never drop a collective in flight
                self._pending[0].wait()
            local = preconditioned
            world = dist.get_world_size()
            gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self._pending = (dist.all_gather_into_tensor(gathered, local, async_op=True), gathered, local)

    def register_hooks(self) -> None:
        module = self.module
        storage = module.storage

        @torch.no_grad()
        def forward_hook(mod: nn.Module, inputs: Tuple[torch.Tensor], outputs: torch.Tensor) -> None:
            del mod
            self._cache_activation(inputs[0].detach())
            self.cached_hooks.append(
                outputs.register_hook(shared_backward_hook if module.factor_args.has_shared_parameters else backward_hook))

        @torch.no_grad()
        def backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            if module.per_sample_gradient_process_fnc is None and module.factor_args.strategy in ("ekfac", "kfac"):
                g, a, ones = module.gradient_factors(activation, output_gradient.detach())
                if self.EIGENBASIS_QUERIES and g.shape[1] == 1 and not module.factor_args.has_shared_parameters:
                    q = g.shape[0]
                    gt = ops.matmul_nn(g.reshape(q, -1), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME))
                    at = ops.matmul_nn(a.reshape(q, -1), self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), append_ones=ones)
                    o, ip = gt.shape[1], at.shape[1]
                    rotated = torch.empty((q, o, ip), dtype=torch.float32, device=g.device)
                    ops.gemm(rotated, ip, o * ip, ops.view(gt, o, 1, o, o, 1), ops.view(at, ip, 1, ip, ip, 1), batch=q,
                             alpha=module.gradient_scale, mul=storage[LAMBDA_MATRIX_NAME])
                    module.queries_in_eigenbasis = True
                    module.query_padding = 0
                    self._store(rotated, from_hook=True)
                    return
                module.queries_in_eigenbasis = False
                qa16, qgt16, qat16, qg16, bias_row = self._bf16_eigenvectors()
                if qa16 is not None and ops.precondition_bf16_eligible(g, a) and (not ones or bias_row is not None):
                    # bf16 eigenvectors only (kf_precondition_bf16): no fp32 copies of Q_G / Q_A are made for this layer
                    out = ops.precondition_bf16(g, a, ones, qg16, qgt16, qa16, qat16, bias_row, storage[LAMBDA_MATRIX_NAME],
                                                scale=module.gradient_scale)
                else:
                    out = ops.precondition(g, a, ones, self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME),
                                           self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), storage[LAMBDA_MATRIX_NAME],
                                           scale=module.gradient_scale, out_dtype=self._out_dtype(),
                                           q_a_bf16=qa16, q_g_t_bf16=qgt16, q_a_t_bf16=qat16)
                # the bf16 engine hands back rows zero-padded to a multiple of 8 (odd I'); the score trackers consume that
                # width as it is, every other reader strips it (``unpadded_queries``)
                module.query_padding = out.shape[-1] - (a.shape[-1] + int(ones))
                self._store(out, from_hook=True)
            else:
                module.queries_in_eigenbasis = False
                module.query_padding = 0
                psg = module.compute_per_sample_gradient(activation, output_gradient.detach())
                out = FactorConfig.CONFIGS[module.factor_args.strategy].precondition_gradient(psg, storage)
                if module.gradient_scale != 1.0:
                    out.mul_(module.gradient_scale)
                self._store(out, from_hook=True)

        @torch.no_grad()
        def shared_backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            psg = module.compute_per_sample_gradient(activation, output_gradient.detach())
            if self.cached_per_sample_gradient is None:
                self.cached_per_sample_gradient = torch.zeros_like(psg)
            self.cached_per_sample_gradient.add_(psg)

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    @torch.no_grad()
    def finalize_iteration(self) -> None:
        module = self.module
        if module.factor_args.has_shared_parameters and self.cached_per_sample_gradient is not None:
            module.queries_in_eigenbasis = False
            module.query_padding = 0
            out = FactorConfig.CONFIGS[module.factor_args.strategy].precondition_gradient(
                self.cached_per_sample_gradient, module.storage)
            if module.gradient_scale != 1.0:
                out.mul_(module.gradient_scale)
            self._store(out)
        self.clear_all_cache()

    def exist(self) -> bool:
        storage = self.module.storage
        return (storage[PRECONDITIONED_GRADIENT_NAME] is not None
                or storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] is not None)

    @staticmethod
    def _interleave(gathered: torch.Tensor, q: int, num_processes: int) -> torch.Tensor:
        """rank-major ``[P * q, ...]`` -> row ``j * P + r`` = rank ``r``'s ``j``-th query."""
        stacked = gathered.reshape((num_processes, q) + tuple(gathered.shape[1:]))
        return stacked.transpose(0, 1).reshape((num_processes * q,) + tuple(gathered.shape[1:]))

    @classmethod
    def _gather_interleaved(cls, local: torch.Tensor, num_processes: int) -> torch.Tensor:
        local = local.contiguous()
        q = local.shape[0]
        gathered = torch.empty((num_processes * q,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        with exchange("query_all_gather", gathered.numel() * gathered.element_size()):
            dist.all_gather_into_tensor(gathered, local)  # rank-major concatenation (layout both RCCL and gloo accept)
        return cls._interleave(gathered, q, num_processes)

    def synchronize(self, num_processes: int = 1) -> None:
        """C4: all-gather the ``[q, O, I']`` block (or both low-rank factors) of every rank and interleave so that
        row ``j * P + r`` is rank ``r``'s ``j``-th query -- the dataset order of a strided ``DistributedSampler``
        (reference ``precondition.py:181-201``)."""
        storage = self.module.storage
        local = storage[PRECONDITIONED_GRADIENT_NAME]
        pending, self._pending = self._pending, None
        if not dist.is_initialized() or local is None:
            return
        if pending is not None and pending[2] is local:  # issued from the backward hook: only the exposed wait is left
            work, gathered, _ = pending
            with exchange("query_all_gather", gathered.numel() * gathered.element_size()):
                work.wait()
            storage[PRECONDITIONED_GRADIENT_NAME] = self._interleave(gathered, local.shape[0], num_processes)
            return
        if pending is not None:
            pending[0].wait()
        if isinstance(local, list):
            storage[PRECONDITIONED_GRADIENT_NAME] = [self._gather_interleaved(t, num_processes) for t in local]
        else:
            storage[PRECONDITIONED_GRADIENT_NAME] = self._gather_interleaved(local, num_processes)

    def truncate(self, keep_size: int) -> None:
        storage = self.module.storage
        held = storage[PRECONDITIONED_GRADIENT_NAME]
        if isinstance(held, list):
            storage[PRECONDITIONED_GRADIENT_NAME] = [t[:keep_size].clone() for t in held]
        else:
            storage[PRECONDITIONED_GRADIENT_NAME] = held[:keep_size].clone()

    def accumulate_iterations(self) -> None:
        storage = self.module.storage
        held, new = storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME], storage[PRECONDITIONED_GRADIENT_NAME]
        if new is None:
            return
        if isinstance(new, list):
            storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = (
                [t.contiguous() for t in new] if held is None
                else [torch.cat((h, t), dim=0).contiguous() for h, t in zip(held, new)])
        elif held is None:  # dense blocks are collected, not concatenated (see QueryBlocks / QueryBuffer)
            capacity = self.module.query_capacity
            storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = (
                QueryBuffer(new, capacity) if capacity and capacity >= new.shape[0] else QueryBlocks([new.contiguous()]))
            self._held_layout = (tuple(new.shape[1:]), self.module.query_padding, self.module.queries_in_eigenbasis)
        else:
            # ``query_padding`` / ``queries_in_eigenbasis`` are per-module flags describing ALL held blocks: a later query
            # batch laid out differently (a one-row batch after sequence batches, say) is brought to the held layout or refused
            layout = (tuple(new.shape[1:]), self.module.query_padding, self.module.queries_in_eigenbasis)
            if layout != self._held_layout:
                shape, padding, eigen = self._held_layout
                if tuple(new.shape[1:-1]) != shape[:-1] or new.shape[-1] - layout[1] != shape[-1] - padding:
                    raise RuntimeError(
                        f"Module '{self.module.name}': query batches produced preconditioned gradients of different shapes "
                        f"({self._held_layout} then {layout}).")
                # a later query batch laid out differently (a one-row batch among sequence batches: eigenbasis-resident,
                # unpadded fp32 against padded bf16 parameter-space blocks, or the other way round) is brought to the held
                # layout -- the reference accepts such mixes (precondition.py:203-240)
                new = new[..., :new.shape[-1] - layout[1]]
                if eigen != layout[2]:
                    new = self._change_basis(new, into_eigenbasis=eigen)
                if new.dtype != held.dtype:
                    new = new.to(held.dtype)
                new = torch.nn.functional.pad(new, (0, padding))
                self.module.query_padding, self.module.queries_in_eigenbasis = padding, eigen
            held.append(new.contiguous())
        storage[PRECONDITIONED_GRADIENT_NAME] = None

    def _change_basis(self, block: torch.Tensor, into_eigenbasis: bool) -> torch.Tensor:
        """``Q_G^T P Q_A`` (parameter space -> eigenbasis) or ``Q_G M Q_A^T`` (back) for a ``[q, O, I']`` block, fp32."""
        storage = self.module.storage
        q_a = storage[ACTIVATION_EIGENVECTORS_NAME].to(dtype=torch.float32).contiguous()
        q_g = storage[GRADIENT_EIGENVECTORS_NAME].to(dtype=torch.float32).contiguous()
        block = block.to(torch.float32).contiguous()
        q, o, ip = block.shape
        t1 = torch.empty((q * o, ip), dtype=torch.float32, device=block.device)
        out = torch.empty((q, o, ip), dtype=torch.float32, device=block.device)
        if into_eigenbasis:
            ops.gemm(t1, ip, 0, ops.view(block, 0, ip, 1, q * o, ip), ops.view(q_a, 0, 1, ip, ip, ip))
            ops.gemm(out, ip, o * ip, ops.view(q_g, 0, 1, o, o, o), ops.view(t1, o * ip, 1, ip, ip, o), batch=q)
        else:
            ops.gemm(t1, ip, 0, ops.view(block, 0, ip, 1, q * o, ip), ops.view(q_a, 0, ip, 1, ip, ip))
            ops.gemm(out, ip, o * ip, ops.view(q_g, 0, o, 1, o, o), ops.view(t1, o * ip, 1, ip, ip, o), batch=q)
        return out

    @torch.no_grad()
    def finalize_all_iterations(self) -> None:
        """``aggregate_query_gradients``: preconditions the summed query gradient held by the ``GradientTracker``
        and makes it the (single-row) accumulated query gradient (reference ``precondition.py:242-255``)."""
        storage = self.module.storage
        summed = storage[AGGREGATED_GRADIENT_NAME]
        if summed is None:
            return
        self.module.queries_in_eigenbasis = False
        self.module.query_padding = 0
        out = FactorConfig.CONFIGS[self.module.factor_args.strategy].precondition_gradient(
            summed.to(torch.float32).contiguous(), storage)
        storage[AGGREGATED_GRADIENT_NAME] = None
        self._store(out if out is not summed else out.clone())
        self.accumulate_iterations()

    def release_memory(self) -> None:
        if self._pending is not None:
            self._pending[0].wait()
            self._pending = None
        self._bf16_q = None
        self._held_layout = None
        self.module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = None
        self.module.storage[PRECONDITIONED_GRADIENT_NAME] = None
        self.clear_all_cache()
