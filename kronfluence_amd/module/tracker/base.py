"""Tracker base: the per-``ModuleMode`` hook logic of a ``TrackedModule``.

One tracker object exists per (module, mode).  The protocol -- ``register_hooks`` ... ``release_memory`` -- carries the
reference's method names (``module/tracker/base.py:8-88``) because ``TrackedModule`` and the model-wide helpers of
``module/utils.py`` dispatch on them; everything else here (the activation stack shared by the forward and backward
hooks of one iteration) is this engine's own plumbing.
"""

from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch.utils.hooks import RemovableHandle


class QueryBlocks:
    """The preconditioned query gradients of several query batches, kept as the list of per-batch ``[q_i, O, I']`` blocks.
    The reference grows one tensor with ``torch.cat`` per batch (``tracker/precondition.py:216-240``) -- quadratic copying,
    2.9 TB for the 174 GB of a GPT-2-small query set; the score trackers consume the blocks as they are (the k-tile-major
    layout is built block by block), so nothing is ever concatenated on the hot path."""

    def __init__(self, blocks) -> None:
        self.blocks = list(blocks)

    def append(self, block: torch.Tensor) -> "QueryBlocks":
        self.blocks.append(block)
        return self

    @property
    def shape(self):
        return (sum(b.shape[0] for b in self.blocks),) + tuple(self.blocks[0].shape[1:])

    @property
    def dtype(self):
        return self.blocks[0].dtype

    @property
    def device(self):
        return self.blocks[0].device

    def dense(self) -> torch.Tensor:
        return self.blocks[0] if len(self.blocks) == 1 else torch.cat(self.blocks, dim=0)


class QueryBuffer(QueryBlocks):
    """``QueryBlocks`` over ONE preallocated ``[capacity, O, I']`` buffer that the query batches are copied into: the stage
    loop knows how many queries it is going to hold (``TrackedModule.query_capacity``), so each layer makes a single
    allocation instead of one per batch -- at BERT / GPT-2 scale (150-175 GB of query gradients, 32 batches x 48 layers)
    the per-batch blocks fragmented the caching allocator into an out-of-memory with 100 GB "reserved but unallocated"."""

    def __init__(self, first: torch.Tensor, capacity: int) -> None:
        self.buffer = torch.empty((capacity,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        self.filled = 0
        self.overflow = []
        self.append(first)

    def append(self, block: torch.Tensor) -> "QueryBuffer":
        q = block.shape[0]
        if not self.overflow and self.filled + q <= self.buffer.shape[0]:
            self.buffer[self.filled:self.filled + q].copy_(block)
            self.filled += q
        else:  # more queries than announced: keep them as extra blocks
            self.overflow.append(block.contiguous())
        return self

    @property
    def blocks(self):
        return [self.buffer[:self.filled]] + self.overflow


def _remove_all(handles: List[RemovableHandle]) -> List[RemovableHandle]:
    for handle in reversed(handles):
        handle.remove()
    return []


class BaseTracker:
    def __init__(self, module: "torch.nn.Module") -> None:
        self.module = module
        self.registered_hooks: List[RemovableHandle] = []   # forward hooks, alive while the mode is active
        self.cached_hooks: List[RemovableHandle] = []       # tensor (backward) hooks of the current iteration
        self.cached_activations: Optional[Union[List[torch.Tensor], torch.Tensor]] = None
        self.cached_per_sample_gradient: Optional[torch.Tensor] = None

    # -- protocol, overridden per mode ---------------------------------------------------------------------
    def register_hooks(self) -> None:
        """Install the mode's forward hook(s) on the wrapped module."""

    def finalize_iteration(self) -> None:
        """After one forward/backward: consume what the hooks accumulated (shared-parameter sums)."""

    def exist(self) -> bool:
        """Whether the mode's result is present in ``module.storage``."""
        return False

    def synchronize(self, num_processes: int) -> None:
        """Exchange the mode's result between ranks."""

    def truncate(self, keep_size: int) -> None:
        """Drop the wrap-around padding rows of the last distributed query batch."""

    def accumulate_iterations(self) -> None:
        """Move one iteration's result into the multi-iteration accumulator."""

    def finalize_all_iterations(self) -> None:
        """After the last iteration of a pass."""

    def release_memory(self) -> None:
        """Free everything the mode keeps in ``module.storage``."""

    # -- shared plumbing ---------------------------------------------------------------------------------------
    def release_hooks(self) -> None:
        self.clear_all_cache()
        self.registered_hooks = _remove_all(self.registered_hooks)

    def clear_all_cache(self) -> None:
        self.cached_activations = None
        self.cached_per_sample_gradient = None
        self.cached_hooks = _remove_all(self.cached_hooks)

    def _raise_cache_not_found_exception(self) -> None:
        raise RuntimeError(
            f"No cached activation for module '{self.module.name}' when its backward hook fired. Either the module "
            "did not take part in the forward pass, or it ran more than once per forward pass -- in that case enable "
            "`FactorArguments.has_shared_parameters` so that activations are stacked per use."
        )

    def _cache_activation(self, activation: torch.Tensor) -> None:
        """Forward hook side: one slot, or a LIFO stack when the module's parameters are shared between uses.
        The hooked input is held by reference (``.detach()``, no copy -- reference ``tracker/pairwise_score.py:58``)
        together with its version counter: all parameters are frozen, so autograd would NOT notice a later in-place
        write to this tensor; ``_take_activation`` does, and raises instead of scoring a corrupted activation."""
        entry = (activation, activation._version)
        if not self.module.factor_args.has_shared_parameters:
            self.cached_activations = entry
        elif self.cached_activations is None:
            self.cached_activations = [entry]
        else:
            self.cached_activations.append(entry)

    def _take_activation(self) -> torch.Tensor:
        """Backward hook side: the activation of the use whose gradient just arrived (last in, first out)."""
        held = self.cached_activations
        entry = None
        if isinstance(held, list):
            if held:
                entry = held.pop()
        elif held is not None:
            self.cached_activations = None
            entry = held
        if entry is None:
            self._raise_cache_not_found_exception()
        activation, version = entry
        if activation._version != version:
            raise RuntimeError(
                f"The input of module '{self.module.name}' was modified in place after its forward pass; the influence "
                "hooks hold that tensor by reference.  Make the offending operation out-of-place (e.g. `inplace=False`)."
            )
        return activation
