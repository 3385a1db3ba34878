"""Summed-gradient tracker for ``aggregate_query_gradients`` / ``aggregate_train_gradients`` (reference
``module/tracker/gradient.py:11-105``).

``storage["aggregated_gradient"] += sum_b g_b`` is ONE GEMM over all ``b * R`` rows of the hooked factors
(``G^T [A, 1]``, depth ``b R``) accumulated in place in fp32 -- the per-sample gradients are never formed.
"""

from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.module.tracker.base import BaseTracker
from kronfluence_amd.utils.constants import AGGREGATED_GRADIENT_NAME


class GradientTracker(BaseTracker):
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
            if module.per_sample_gradient_process_fnc is None:
                g, a, ones = module.gradient_factors(activation, output_gradient.detach())
                o, ip = g.shape[-1], a.shape[-1] + int(ones)
                if storage[AGGREGATED_GRADIENT_NAME] is None:
                    storage[AGGREGATED_GRADIENT_NAME] = torch.zeros((1, o, ip), dtype=torch.float32, device=g.device)
                module.accumulate_summed_gradient(storage[AGGREGATED_GRADIENT_NAME], g, a, ones, module.gradient_scale)
            else:
                psg = module.compute_per_sample_gradient(activation, output_gradient.detach()).to(torch.float32).contiguous()
                b, o, ip = psg.shape
                if storage[AGGREGATED_GRADIENT_NAME] is None:
                    storage[AGGREGATED_GRADIENT_NAME] = torch.zeros((1, o, ip), dtype=torch.float32, device=psg.device)
                ones_vec = torch.ones(b, dtype=torch.float32, device=psg.device)
                ops.gemm(storage[AGGREGATED_GRADIENT_NAME], o * ip, 0, ops.view(ones_vec, 0, 0, 1, 1, b),
                         ops.view(psg, 0, 1, o * ip, o * ip, b), alpha=module.gradient_scale, beta=1.0)

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    def finalize_iteration(self) -> None:
        self.clear_all_cache()

    def exist(self) -> bool:
        return self.module.storage[AGGREGATED_GRADIENT_NAME] is not None

    def synchronize(self, num_processes: int = 1) -> None:
        """SUM all-reduce of the summed gradient (RCCL on GPU, gloo in the CPU tests)."""
        del num_processes
        if dist.is_initialized() and self.exist():
            dist.all_reduce(self.module.storage[AGGREGATED_GRADIENT_NAME], op=dist.ReduceOp.SUM)

    def release_memory(self) -> None:
        self.clear_all_cache()
        self.module.storage[AGGREGATED_GRADIENT_NAME] = None
