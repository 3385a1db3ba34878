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


def _device_tensors(value, _depth: int = 0):
    """Every GPU tensor reachable from ``value`` through dicts / lists / tuples and the tensor attributes of plain holder
    objects (``TiledQueries`` and the like), two levels deep."""
    if isinstance(value, torch.Tensor):
        if value.is_cuda:
            yield value
    elif isinstance(value, dict):
        for item in value.values():
            yield from _device_tensors(item, _depth)
    elif isinstance(value, (list, tuple)):
        for item in value:
            yield from _device_tensors(item, _depth)
    elif value is not None and _depth < 2 and hasattr(value, "__dict__"):
        yield from _device_tensors(vars(value), _depth + 1)


class BaseTracker:
    def __init__(self, module: "torch.nn.Module") -> None:
        self.module = module
        self.registered_hooks: List[RemovableHandle] = []   # forward hooks, alive while the mode is active
        self.cached_hooks: List[RemovableHandle] = []       # tensor (backward) hooks of the current iteration
        self.cached_activations: Optional[Union[List[torch.Tensor], torch.Tensor]] = None
        self.cached_per_sample_gradient: Optional[torch.Tensor] = None

    def _eigenvectors32(self, name: str) -> torch.Tensor:
        """The stored eigenvector matrix ``name`` in fp32, converted (and kept in ``storage``) on first use: ``Ekfac.prepare`` leaves
        accelerator-resident low-precision eigenvectors as they are when the preconditioner runs in bf16 -- the bf16 call chain
        never reads fp32 copies -- so the paths that do (one row per sample, fp32 stages, self-influence) ask here."""
        q = self.module.storage[name]
        if q.dtype != torch.float32 or not q.is_contiguous():
            q = q.to(dtype=torch.float32).contiguous()
            self.module.storage[name] = q
        return q

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

    # -- a second HIP stream for the hooks' kernels (OPT-IN: KF_SIDE_STREAM=1 or "auto") -----------------------------------
    # The kernels a hook launches depend on nothing the model computes after the hook, so they can run BESIDE the rest of the
    # model's pass (its normalisation / pooling / elementwise kernels leave the matrix cores idle, the EK-FAC GEMMs leave HBM
    # idle) instead of in line with it.  One stream per process, shared by all layers and modes (kernels that accumulate into
    # the same buffers stay ordered among themselves).  Measured on the MI355X (profiles/README.md, round 4): the ResNet-9
    # pairwise stage gains 1.6 - 3.3 % (8.9 GiB in use); BERT-base LOSES 2x and GPT-2-small runs out of memory with 174 - 250
    # GiB in use -- a second stream is a second pool of the caching allocator, the workspaces of the score kernels no longer
    # reuse the blocks of the main pool -- and every overlapped kernel runs slower than alone, so the event-timed rooflines of
    # bench.py no longer describe the kernels.  Hence off by default; "1" takes it always, "auto" only while more than half of
    # the device memory is free at a layer's first hook of a pass (sticky for that pass).  Only the stage loops opt in
    # (``TrackedModule.side_stream_ok``): they join the stream before anyone reads the results.  The hooked tensors are kept
    # referenced until the layer's next call: autograd accumulates later gradient contributions IN PLACE into a buffer it
    # alone owns (the output gradient of a projection feeding a residual sum is that sum's gradient), which would race with
    # kernels that have not run yet.
    _SIDE: dict = {}    # device -> its side stream (a stream is never replaced while events recorded on it are pending)
    _side_done = None   # event: this layer's kernels of the previous hook call on the side stream
    _side_used = None   # the stream that event was recorded on
    _side_keep = None   # the hooked tensors of that call
    _use_side = None    # this layer's decision for the current pass
    SIDE_STREAM_MIN_FREE = 0.5

    def _side_stream(self, device):
        import os

        if device is None or device.type != "cuda" or not getattr(self.module, "side_stream_ok", False):
            return None
        if self._use_side is None:
            mode = os.environ.get("KF_SIDE_STREAM", "0")
            if mode == "auto":
                free, total = torch.cuda.mem_get_info(device)
                self._use_side = free > self.SIDE_STREAM_MIN_FREE * total
            else:
                self._use_side = mode == "1"
        if not self._use_side:
            return None
        key = (device.type, torch.cuda.current_device() if device.index is None else device.index)
        stream = BaseTracker._SIDE.get(key)
        if stream is None:
            stream = BaseTracker._SIDE[key] = torch.cuda.Stream(device=device)
        return stream

    def _run_beside(self, device, hooked, work) -> None:
        """Runs ``work()`` (the kernels of one hook call) on the side stream when this layer uses it, else in line.  Ordering:
        the side stream waits for everything enqueued so far (the hooked tensors, the factors); the main stream waits for this
        layer's PREVIOUS call (at most one call per layer in flight: bounds what the side stream keeps alive); the ``hooked``
        tensors are marked as in use on the side stream so that the allocator does not hand their memory out early."""
        side = self._side_stream(device)
        if side is None:
            work()
            return
        main = torch.cuda.current_stream(device)
        if self._side_done is not None:
            main.wait_event(self._side_done)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            work()
            self._side_done = side.record_event()
        self._side_used = side
        for tensor in hooked:
            if tensor.is_cuda:
                tensor.record_stream(side)
        self._side_keep = tuple(hooked)

    def _join_side(self) -> None:
        """End of a pass: the caller's stream waits for the side stream; the decision is taken anew next pass.  What the hooks
        allocated while the side stream was current (the accumulators a mode creates on first use: covariances, Lambda, score
        blocks) belongs to the side stream's pool of the caching allocator but is read on the caller's stream from here on:
        every tensor the layer keeps in ``module.storage`` is marked as in use there, so that a later free cannot hand its
        memory out while a main-stream reader is still pending."""
        side = self._side_used if self._side_done is not None else None
        if side is not None:
            main = torch.cuda.current_stream(side.device)
            main.wait_stream(side)
            for tensor in _device_tensors(self.module.storage):
                tensor.record_stream(main)
            self._side_done = None
        self._side_used = None
        self._side_keep = None
        self._use_side = None

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

    def _offload_activations(self) -> bool:
        """Whether this stage's arguments ask for cached activations in host memory: the score-stage trackers read
        ``ScoreArguments.offload_activations_to_cpu``, ``LambdaTracker`` overrides this with ``FactorArguments``'."""
        return bool(self.module.score_args.offload_activations_to_cpu)

    def _cache_activation(self, activation: torch.Tensor) -> None:
        """Forward hook side: one slot, or a LIFO stack when the module's parameters are shared between uses.
        The hooked input is held by reference (``.detach()``, no copy -- reference ``tracker/pairwise_score.py:58``)
        together with its version counter: all parameters are frozen, so autograd would NOT notice a later in-place
        write to this tensor; ``_take_activation`` does, and raises instead of scoring a corrupted activation."""
        if self._offload_activations() and activation.device.type != "cpu":
            # ``offload_activations_to_cpu`` (reference tracker/factor.py:239, pairwise_score.py:59, ...): the hooked input waits
            # for its gradient in host memory -- a COPY, so a later in-place write to the original cannot reach it; the
            # entry remembers the device to come back to
            entry = (activation.to("cpu"), None, activation.device)
        else:
            entry = (activation, activation._version, None)
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
        activation, version, home = entry
        if home is not None:
            return activation.to(home)
        if activation._version != version:
            raise RuntimeError(
                f"The input of module '{self.module.name}' was modified in place after its forward pass; the influence "
                "hooks hold that tensor by reference.  Make the offending operation out-of-place (e.g. `inplace=False`)."
            )
        return activation
