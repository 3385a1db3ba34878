"""Self-influence trackers (reference ``module/tracker/self_score.py:32-252``).

``SelfScoreTracker``: ``score_n = <P(g_n), g_n>``.  For EK-FAC / K-FAC that is
``sum_{o,i} (Qg^T g_n Qa)^2_{oi} * Lambda^-1_{oi}``, so the back-rotation of the reference's
``precondition_gradient`` is never needed: the gradient's factors are rotated into the eigenbasis (two GEMMs),
the rotated per-sample gradient is formed once, and ``kf_rowwise_dot`` reduces its weighted square.
``SelfScoreWithMeasurementTracker``: ``score_n = <P(grad measurement_n), grad loss_n>`` -- the first backward runs in
``PRECONDITION_GRADIENT`` mode, the second one here.

Every layer accumulates into one ``[N_shard]`` fp32 vector resident in HBM (``module.score_sink``) instead of a
per-layer vector that the stage loop adds up and copies to the host after every batch (``score/self.py:243-262``).
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.factor.config import FactorConfig
from kronfluence_amd.module.tracker.base import BaseTracker
from kronfluence_amd.module.tracker.pairwise_score import dense_queries, unpadded_queries
from kronfluence_amd.utils.constants import (
    ACTIVATION_EIGENVECTORS_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    LAMBDA_MATRIX_NAME,
    PRECONDITIONED_GRADIENT_NAME,
    SELF_SCORE_VECTOR_NAME,
)


class _SelfScoreBase(BaseTracker):
    def _target(self, batch: int, device: torch.device) -> torch.Tensor:
        """The ``[batch]`` slice of the stage's score vector (or a private vector when the module API is used
        directly); contributions are ADDED to it."""
        module = self.module
        if module.score_sink is not None:
            buffer, offset = module.score_sink
            return buffer[offset:offset + batch]
        held = module.storage[SELF_SCORE_VECTOR_NAME]
        if held is None or held.shape[0] != batch:
            held = torch.zeros(batch, dtype=torch.float32, device=device)
            module.storage[SELF_SCORE_VECTOR_NAME] = held
        return held

    def exist(self) -> bool:
        return self.module.storage[SELF_SCORE_VECTOR_NAME] is not None or self.module.score_sink is not None

    def accumulate_iterations(self) -> None:
        self.clear_all_cache()
        self.module.storage[SELF_SCORE_VECTOR_NAME] = None

    def finalize_all_iterations(self) -> None:
        self.module.score_sink = None
        self.clear_all_cache()

    def release_memory(self) -> None:
        self.clear_all_cache()
        self.module.storage[SELF_SCORE_VECTOR_NAME] = None


class SelfScoreTracker(_SelfScoreBase):
    def _score_from_gradient(self, per_sample_gradient: torch.Tensor) -> None:
        """Generic form on a materialised ``[b, O, I']`` gradient (post-processed / shared-parameter gradients and
        the strategies without an eigenbasis): ``<precondition(g), g>`` (reference ``self_score.py:38-62``)."""
        module = self.module
        psg = per_sample_gradient.contiguous()
        preconditioned = FactorConfig.CONFIGS[module.factor_args.strategy].precondition_gradient(psg, module.storage)
        scale = module.gradient_scale
        ops.rowwise_dot(self._target(psg.shape[0], psg.device), preconditioned, psg, None, scale=scale * scale)

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
                b, r, o = g.shape
                gt = ops.matmul_nn(g.reshape(b * r, o), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME))
                at = ops.matmul_nn(a.reshape(b * r, a.shape[-1]), self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), append_ones=ones)
                ip = at.shape[1]
                rotated = torch.empty((b, o, ip), dtype=torch.float32, device=g.device)
                ops.gemm(rotated, ip, o * ip, ops.view(gt, r * o, 1, o, o, r), ops.view(at, r * ip, 1, ip, ip, r), batch=b,
                         alpha=module.gradient_scale)
                ops.rowwise_dot(self._target(b, g.device), rotated, rotated, storage[LAMBDA_MATRIX_NAME])
            else:
                self._score_from_gradient(module.compute_per_sample_gradient(activation, output_gradient.detach()))

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
        if self.module.factor_args.has_shared_parameters and self.cached_per_sample_gradient is not None:
            self._score_from_gradient(self.cached_per_sample_gradient)
        self.clear_all_cache()


class SelfScoreWithMeasurementTracker(_SelfScoreBase):
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
            preconditioned = storage[PRECONDITIONED_GRADIENT_NAME]
            if preconditioned is None:
                raise RuntimeError(f"Module '{module.name}' holds no preconditioned measurement gradient.")
            preconditioned = dense_queries(unpadded_queries(module, preconditioned))
            if module.per_sample_gradient_process_fnc is None:
                g, a, ones = module.gradient_factors(activation, output_gradient.detach())
                if module.queries_in_eigenbasis:  # see PreconditionTracker.EIGENBASIS_QUERIES
                    # rotated row by row, keeping the R axis: the held queries had one row per sample, the train
                    # batch may have several (single-token queries against sequence batches)
                    n, r = g.shape[0], g.shape[1]
                    g = ops.matmul_nn(g.reshape(n * r, -1), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME)).reshape(n, r, -1)
                    a = ops.matmul_nn(a.reshape(n * r, -1), self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME),
                                      append_ones=ones).reshape(n, r, -1)
                    ones = False
                psg = ops.per_sample_gradient(g, a, ones)
            else:
                psg = module.compute_per_sample_gradient(activation, output_gradient.detach()).contiguous()
            ops.rowwise_dot(self._target(psg.shape[0], psg.device), preconditioned, psg, None, scale=module.gradient_scale)
            if not module.factor_args.has_shared_parameters:
                storage[PRECONDITIONED_GRADIENT_NAME] = None

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    def finalize_iteration(self) -> None:
        self.module.storage[PRECONDITIONED_GRADIENT_NAME] = None
        self.clear_all_cache()
