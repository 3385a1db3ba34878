"""``Task``: the user-supplied description of the training objective and the query measurement.

Same abstract interface as the reference's ``kronfluence/task.py:8-116`` (method names, argument
order and meaning), so an existing ``Task`` subclass works unchanged.
"""

from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Union

import torch
from torch import nn


class Task(ABC):
    """Subclass and implement ``compute_train_loss`` and ``compute_measurement``."""

    enable_post_process_per_sample_gradient: bool = False

    @abstractmethod
    def compute_train_loss(self, batch: Any, model: nn.Module, sample: bool = False) -> torch.Tensor:
        """Summed (not averaged) training loss of ``batch``.  ``sample=True`` must draw the targets
        from the model's own predictive distribution (true Fisher)."""
        raise NotImplementedError(f"{type(self).__name__} must implement `compute_train_loss`.")

    @abstractmethod
    def compute_measurement(self, batch: Any, model: nn.Module) -> torch.Tensor:
        """The query-side quantity f(theta) whose gradient is scored (loss, logit, margin, ...)."""
        raise NotImplementedError(f"{type(self).__name__} must implement `compute_measurement`.")

    def get_influence_tracked_modules(self) -> Optional[List[str]]:
        """Names of the modules to track; ``None`` tracks every supported leaf (Linear, Conv2d)."""
        return None

    def get_attention_mask(self, batch: Any) -> Optional[Union[Dict[str, torch.Tensor], torch.Tensor]]:
        """Binary padding mask ``[batch, seq]`` (or a per-module dict) used by the covariance stage."""
        return None

    def post_process_per_sample_gradient(self, module_name: str, gradient: torch.Tensor) -> torch.Tensor:
        """Hook to edit a module's per-sample gradient ``[batch, out, in]``; used only when
        ``enable_post_process_per_sample_gradient`` is set."""
        del module_name
        return gradient
