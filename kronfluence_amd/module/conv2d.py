"""``TrackedConv2d``: per-module operators for ``nn.Conv2d`` on the HIP kernels
(reference ``module/conv2d.py:67-227``).

Patches are produced by ``kf_im2col`` (group-mean + unfold + ones column in one pass); the output
gradient is consumed in its native NCHW layout for the covariance and as ``[b, P, O]`` rows for the
per-sample-gradient contractions.
"""

from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracked_module import TrackedModule


class TrackedConv2d(TrackedModule, module_type=nn.Conv2d):
    @property
    def in_channels(self) -> int:
        return self.original_module.in_channels

    @property
    def out_channels(self) -> int:
        return self.original_module.out_channels

    @property
    def kernel_size(self) -> Tuple[int, int]:
        return self.original_module.kernel_size

    @property
    def padding(self) -> Tuple[int, int]:
        return self.original_module.padding

    @property
    def dilation(self) -> Tuple[int, int]:
        return self.original_module.dilation

    @property
    def groups(self) -> int:
        return self.original_module.groups

    @property
    def padding_mode(self) -> str:
        return self.original_module.padding_mode

    @property
    def weight(self) -> torch.Tensor:
        return self.original_module.weight

    @property
    def bias(self) -> Optional[torch.Tensor]:
        return self.original_module.bias

    def _patches(self, x: torch.Tensor) -> torch.Tensor:
        dtype = x.dtype if x.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
        return ops.im2col(x, self.original_module, self.has_bias, dtype)

    # -- fused hot path ----------------------------------------------------------------------------
    def accumulate_activation_covariance(self, cov, count, input_activation):
        geometry = ops.conv2d_cov_geometry(input_activation, self.original_module)
        if geometry is not None:  # implicit im2col: no patch tensor (kf_conv2d_cov_accum)
            d = input_activation.shape[1] * self.kernel_size[0] * self.kernel_size[1]
            if cov is None:
                cov = torch.zeros((d, d), dtype=torch.float32, device=input_activation.device)
                count = torch.zeros(1, dtype=torch.int64, device=input_activation.device)
            ops.conv2d_cov_accum(cov, count, input_activation, self.original_module, geometry)
            return cov, count
        conv = self.original_module
        d_small = (input_activation.shape[1] * self.kernel_size[0] * self.kernel_size[1] + int(self.has_bias)
                   if (input_activation.dim() == 4 and conv.groups == 1) else 33)
        if d_small <= 32:  # a first layer's few patch columns: one fp32 MFMA per two positions, no patch tensor (kf_conv2d_cov_small)
            fresh = cov is None
            if fresh:
                cov = torch.zeros((d_small, d_small), dtype=torch.float32, device=input_activation.device)
                count = torch.zeros(1, dtype=torch.int64, device=input_activation.device)
            if ops.conv2d_cov_small(cov, count, input_activation, conv):
                return cov, count
            if fresh:
                cov = count = None
        d_rows = (input_activation.shape[1] * self.kernel_size[0] * self.kernel_size[1] + int(self.has_bias)
                  if (input_activation.dim() == 4 and conv.groups == 1) else 0)
        if d_rows >= 256 and d_rows % 8 == 0:   # an output grid that is not whole k-steps: patch rows on the K-major covariance kernel
            fresh = cov is None
            if fresh:
                cov = torch.zeros((d_rows, d_rows), dtype=torch.float32, device=input_activation.device)
                count = torch.zeros(1, dtype=torch.int64, device=input_activation.device)
            if ops.conv_patch_rows_cov(cov, count, input_activation, conv):
                return cov, count
            if fresh:
                cov = count = None
        patches = self._patches(input_activation)
        d = patches.shape[-1]
        if cov is None:
            cov = torch.zeros((d, d), dtype=torch.float32, device=patches.device)
            count = torch.zeros(1, dtype=torch.int64, device=patches.device)
        n = patches.shape[0] * patches.shape[1]
        ops.syrk_accum(cov, patches, n, d, max(n, 1), 0, d, 1, None, False, 1.0, count)
        return cov, count

    def accumulate_gradient_covariance(self, cov, count, output_gradient, alpha):
        d = output_gradient.shape[1]
        if cov is None:
            cov = torch.zeros((d, d), dtype=torch.float32, device=output_gradient.device)
            count = torch.zeros(1, dtype=torch.int64, device=output_gradient.device)
        ops.conv_gradient_cov(cov, count, output_gradient, alpha)
        return cov, count

    def gradient_factors(self, input_activation, output_gradient):
        if input_activation.dtype != output_gradient.dtype:
            # as TrackedLinear: both factors in the gradient's dtype (under autocast the first layer sees an fp32 image
            # but computes -- and back-propagates -- in bf16)
            input_activation = input_activation.to(output_gradient.dtype)
        patches = self._patches(input_activation)  # [b, P, I'] incl. the ones column
        grads = output_gradient.flatten(2).transpose(1, 2).contiguous().to(patches.dtype)  # [b, P, O]
        return grads, patches, False

    # -- reference operator API --------------------------------------------------------------------
    def get_flattened_activation(self, input_activation: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        patches = self._patches(input_activation)
        flat = patches.reshape(-1, patches.shape[-1])
        return flat, flat.shape[0]

    def get_flattened_gradient(self, output_gradient: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        flat = output_gradient.permute(0, 2, 3, 1).reshape(-1, output_gradient.shape[1])
        return flat, flat.shape[0]

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
