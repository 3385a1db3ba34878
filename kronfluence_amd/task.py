"""``Task``: the user-supplied description of the training objective and the query measurement.

The interface -- method names, argument order and meaning -- is the reference's ``kronfluence/task.py:8-116``, so an
existing ``Task`` subclass works unchanged: two required methods (the loss the factors are fitted on and the
measurement whose gradient is scored) and three optional hooks.
"""

from typing import Any, Dict, List, Optional, Union

import torch
from torch import nn

MaskType = Optional[Union[Dict[str, torch.Tensor], torch.Tensor]]


def _required(owner: "Task", method: str) -> NotImplementedError:
    return NotImplementedError(f"{type(owner).__name__} must implement `{method}`.")


class Task:
    """Subclass and implement ``compute_train_loss`` and ``compute_measurement``."""

    # set to True to have ``post_process_per_sample_gradient`` applied to every tracked module's gradient
    enable_post_process_per_sample_gradient: bool = False

    # -- optional hooks -----------------------------------------------------------------------------------
    def get_influence_tracked_modules(self) -> Optional[List[str]]:
        """Names of the modules to track; ``None`` tracks every supported leaf (Linear, Conv2d)."""
        return None

    def get_attention_mask(self, batch: Any) -> MaskType:
        """Binary padding mask ``[batch, seq]`` (or a per-module dict) used by the covariance stage."""
        del batch
        return None

    def post_process_per_sample_gradient(self, module_name: str, gradient: torch.Tensor) -> torch.Tensor:
        """Hook to edit a module's per-sample gradient ``[batch, out, in]``; used only when
        ``enable_post_process_per_sample_gradient`` is set."""
        del module_name
        return gradient

    # -- required ---------------------------------------------------------------------------------------------
    def compute_measurement(self, batch: Any, model: nn.Module) -> torch.Tensor:
        """The query-side quantity f(theta) whose gradient is scored (loss, logit, margin, ...)."""
        raise _required(self, "compute_measurement")

    def compute_train_loss(self, batch: Any, model: nn.Module, sample: bool = False) -> torch.Tensor:
        """Summed (not averaged) training loss of ``batch``.  ``sample=True`` must draw the targets
        from the model's own predictive distribution (true Fisher)."""
        raise _required(self, "compute_train_loss")
