"""``TrackedLinear``: per-module operators for ``nn.Linear`` on the HIP kernels
(reference ``module/linear.py:11-138``)."""

from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracked_module import TrackedModule


def _rows(t: torch.Tensor) -> torch.Tensor:
    """``[b, ..., d] -> [b, R, d]`` (R = 1 for 2-D inputs) without copying."""
    t = t if t.is_contiguous() else t.contiguous()
    return t.reshape(t.shape[0], -1, t.shape[-1])


class TrackedLinear(TrackedModule, module_type=nn.Linear):
    @property
    def in_features(self) -> int:
        return self.original_module.in_features

    @property
    def out_features(self) -> int:
        return self.original_module.out_features

    @property
    def weight(self) -> torch.Tensor:
        return self.original_module.weight

    @property
    def bias(self) -> Optional[torch.Tensor]:
        return self.original_module.bias

    def _mask_for(self, n_rows: int) -> Optional[torch.Tensor]:
        mask = self.attention_mask
        return mask if mask is not None and mask.numel() == n_rows else None

    # -- fused hot path ----------------------------------------------------------------------------
    def accumulate_activation_covariance(self, cov, count, input_activation):
        d = input_activation.shape[-1] + int(self.has_bias)
        if cov is None:
            cov = torch.zeros((d, d), dtype=torch.float32, device=input_activation.device)
            count = torch.zeros(1, dtype=torch.int64, device=input_activation.device)
        n = input_activation.numel() // input_activation.shape[-1]
        ops.linear_activation_cov(cov, count, input_activation, self._mask_for(n), self.has_bias)
        return cov, count

    def accumulate_gradient_covariance(self, cov, count, output_gradient, alpha):
        d = output_gradient.shape[-1]
        if cov is None:
            cov = torch.zeros((d, d), dtype=torch.float32, device=output_gradient.device)
            count = torch.zeros(1, dtype=torch.int64, device=output_gradient.device)
        n = output_gradient.numel() // d
        ops.linear_gradient_cov(cov, count, output_gradient, self._mask_for(n), alpha)
        return cov, count

    def gradient_factors(self, input_activation, output_gradient):
        if input_activation.dtype != output_gradient.dtype:
            input_activation = input_activation.to(output_gradient.dtype)
        return _rows(output_gradient), _rows(input_activation), self.has_bias

    # -- reference operator API --------------------------------------------------------------------
    def get_flattened_activation(self, input_activation: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        """API-compatibility form of reference ``linear.py:30-46`` (materialises ``X'``; the covariance
        tracker uses the fused kernel instead)."""
        flat = input_activation.reshape(-1, input_activation.shape[-1])
        mask = self._mask_for(flat.shape[0])
        if mask is not None:
            flat = flat * mask.reshape(-1, 1).to(flat.dtype)
        if self.has_bias:
            ones = flat.new_ones((flat.shape[0], 1))
            if mask is not None:
                ones = ones * mask.reshape(-1, 1).to(flat.dtype)
            flat = torch.cat([flat, ones], dim=-1)
        return flat, (flat.shape[0] if mask is None else mask.sum())

    def get_flattened_gradient(self, output_gradient: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        flat = output_gradient.reshape(-1, output_gradient.shape[-1])
        mask = self._mask_for(flat.shape[0])
        return flat, (flat.shape[0] if mask is None else mask.sum())

    def compute_per_sample_gradient(self, input_activation: torch.Tensor, output_gradient: torch.Tensor) -> torch.Tensor:
        g, a, ones = self.gradient_factors(input_activation, output_gradient)
        per_sample_gradient = ops.per_sample_gradient(g, a, ones)
        if self.per_sample_gradient_process_fnc is not None:
            per_sample_gradient = self.per_sample_gradient_process_fnc(module_name=self.name, gradient=per_sample_gradient)
        return per_sample_gradient

    def compute_pairwise_score(self, preconditioned_gradient: torch.Tensor, input_activation: torch.Tensor,
                               output_gradient: torch.Tensor) -> torch.Tensor:
        g, a, ones = self.gradient_factors(input_activation, output_gradient)
        p = preconditioned_gradient.contiguous()
        if p.dtype not in (torch.float32, torch.bfloat16):
            p = p.to(torch.float32)
        scores = torch.zeros((p.shape[0], g.shape[0]), dtype=torch.float32, device=g.device)
        ops.pairwise_score(scores, 0, p, g, a, ones)
        return scores
