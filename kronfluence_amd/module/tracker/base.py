"""Tracker base: per-``ModuleMode`` hook logic of a ``TrackedModule`` (reference ``tracker/base.py``)."""

from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch.utils.hooks import RemovableHandle


class BaseTracker:
    def __init__(self, module: "torch.nn.Module") -> None:
        self.module = module
        self.registered_hooks: List[RemovableHandle] = []
        self.cached_hooks: List[RemovableHandle] = []
        self.cached_activations: Optional[Union[List[torch.Tensor], torch.Tensor]] = None
        self.cached_per_sample_gradient: Optional[torch.Tensor] = None

    def release_hooks(self) -> None:
        self.clear_all_cache()
        for handle in reversed(self.registered_hooks):
            handle.remove()
        self.registered_hooks = []

    def clear_all_cache(self) -> None:
        self.cached_activations, self.cached_per_sample_gradient = None, None
        for handle in reversed(self.cached_hooks):
            handle.remove()
        self.cached_hooks = []

    def _raise_cache_not_found_exception(self) -> None:
        raise RuntimeError(
            f"Module '{self.module.name}' has no cached activations. This can occur if:\n"
            f"1. The module was not used during the forward pass, or\n"
            f"2. The module was encountered multiple times in the forward pass.\n"
            f"For case 2, set 'has_shared_parameters=True' to enable parameter sharing."
        )

    def _take_activation(self) -> torch.Tensor:
        """Pops the activation cached by the forward hook (LIFO when parameters are shared)."""
        if self.cached_activations is None:
            self._raise_cache_not_found_exception()
        if isinstance(self.cached_activations, list):
            if not self.cached_activations:
                self._raise_cache_not_found_exception()
            return self.cached_activations.pop()
        activation, self.cached_activations = self.cached_activations, None
        return activation

    def _cache_activation(self, activation: torch.Tensor) -> None:
        if self.module.factor_args.has_shared_parameters:
            if self.cached_activations is None:
                self.cached_activations = []
            self.cached_activations.append(activation)
        else:
            self.cached_activations = activation

    # -- overridable protocol (names as in the reference) ---------------------------------------
    def register_hooks(self) -> None: ...
    def finalize_iteration(self) -> None: ...
    def exist(self) -> bool:
        return False
    def synchronize(self, num_processes: int) -> None: ...
    def truncate(self, keep_size: int) -> None: ...
    def accumulate_iterations(self) -> None: ...
    def finalize_all_iterations(self) -> None: ...
    def release_memory(self) -> None: ...
