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
from kronfluence_amd.module.tracker.base import BaseTracker
from kronfluence_amd.utils.constants import (
    ACCUMULATED_PRECONDITIONED_GRADIENT_NAME,
    ACTIVATION_EIGENVECTORS_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    PAIRWISE_SCORE_MATRIX_NAME,
    PRECONDITIONED_GRADIENT_NAME,
)


class PairwiseScoreTracker(BaseTracker):
    _tiled = None  # (source tensor, k-tile-major bf16 copy) of the held query gradients

    def _tiled_queries(self, preconditioned: torch.Tensor, g: torch.Tensor, a: torch.Tensor, ones: bool):
        """k-tile-major copy of the bf16 query gradients, built once per train pass (see
        ``ops.k_tile_major``); ``None`` when the fast layout does not apply."""
        d = preconditioned.shape[1] * preconditioned.shape[2]
        if preconditioned.dtype != torch.bfloat16 or g.shape[1] == 1 or d % 64 != 0 or g.dtype != torch.bfloat16:
            return None
        if self._tiled is None or self._tiled[0] is not preconditioned:
            self._tiled = (preconditioned, ops.k_tile_major(preconditioned))
        return self._tiled[1]

    def register_hooks(self) -> None:
        module = self.module
        storage = module.storage

        @torch.no_grad()
        def forward_hook(mod: nn.Module, inputs: Tuple[torch.Tensor], outputs: torch.Tensor) -> None:
            del mod
            self._cache_activation(inputs[0].detach().clone())
            self.cached_hooks.append(outputs.register_hook(backward_hook))

        @torch.no_grad()
        def backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            preconditioned = storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME]
            if preconditioned is None:
                raise RuntimeError(f"Module '{module.name}' holds no preconditioned query gradient.")
            if preconditioned.dtype not in (torch.float32, torch.bfloat16):
                preconditioned = preconditioned.to(torch.float32)
            batch = output_gradient.shape[0]
            if module.score_sink is not None:
                scores, offset = module.score_sink
            else:
                scores = torch.zeros((preconditioned.shape[0], batch), dtype=torch.float32, device=output_gradient.device)
                offset = 0
                accumulate_into = storage[PAIRWISE_SCORE_MATRIX_NAME] if module.factor_args.has_shared_parameters else None
                if accumulate_into is not None and accumulate_into.shape == scores.shape:
                    scores = accumulate_into
                storage[PAIRWISE_SCORE_MATRIX_NAME] = scores
            if module.per_sample_gradient_process_fnc is None:
                g, a, ones = module.gradient_factors(activation, output_gradient.detach())
                if module.queries_in_eigenbasis:  # see PreconditionTracker.EIGENBASIS_QUERIES
                    n = g.shape[0]
                    g = ops.matmul_nn(g.reshape(n, -1), storage[GRADIENT_EIGENVECTORS_NAME]).unsqueeze(1)
                    a = ops.matmul_nn(a.reshape(n, -1), storage[ACTIVATION_EIGENVECTORS_NAME], append_ones=ones).unsqueeze(1)
                    ones = False
                ops.pairwise_score(scores, offset, preconditioned, g, a, ones, scale=module.gradient_scale,
                                   p_tiled=self._tiled_queries(preconditioned, g, a, ones))
            else:
                # post-processed gradient (pairwise_score.py:41-45): contract the materialised gradient
                psg = module.compute_per_sample_gradient(activation, output_gradient.detach()).contiguous()
                preconditioned = preconditioned.to(torch.float32)
                b, o, ip = psg.shape
                q = preconditioned.shape[0]
                ops.gemm(scores[:, offset:offset + b], scores.shape[1], 0,
                         ops.view(preconditioned.contiguous(), 0, o * ip, 1, q, o * ip),
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
        self.module.storage[ACCUMULATED_PRECONDITIONED_GRADIENT_NAME] = None
        self.module.storage[PRECONDITIONED_GRADIENT_NAME] = None
        self.module.score_sink = None
        self._tiled = None
        self.clear_all_cache()

    def release_memory(self) -> None:
        self.clear_all_cache()
        self.module.storage[PAIRWISE_SCORE_MATRIX_NAME] = None
