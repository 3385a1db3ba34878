"""Process / device state: one process per MI355X, ``torch.distributed`` over RCCL (backend string
"nccl" on ROCm) when launched by ``torchrun``.  Counterpart of the reference's ``utils/state.py``
(rank, device, barrier) without the accelerate dependency."""

from __future__ import annotations

import contextlib
import gc
import os
from typing import Optional

import torch
import torch.distributed as dist


def force_exchanges() -> bool:
    """``KF_DIST_FORCE=1`` (test hook): a ONE-rank process group still runs every exchange of the sharded path -- factor
    all-reduce, eigendecomposition broadcasts, query all-gather, score gather, barriers -- so that the collective library
    (RCCL on the GPU box, which has a single GPU) executes the very calls a multi-rank job makes; with one rank they are
    identities and the results must equal the plain single-process run."""
    return os.environ.get("KF_DIST_FORCE", "0") == "1"


class State:
    """Shared (Borg) view of the launch environment."""

    _shared: dict = {}

    def __init__(self, cpu: bool = False) -> None:
        self.__dict__ = State._shared
        if getattr(self, "initialized", False):
            if cpu and self.device.type != "cpu":
                raise ValueError("State was already initialised on a GPU; cannot switch to `cpu=True`.")
            return
        self.cpu = cpu
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "-1"))
        if dist.is_available() and dist.is_initialized():
            self.num_processes, self.process_index = dist.get_world_size(), dist.get_rank()
            self.local_process_index = max(local_rank, 0)
        elif (world > 1 or force_exchanges()) and local_rank >= 0:
            backend = "gloo" if cpu or not torch.cuda.is_available() else "nccl"  # "nccl" == RCCL on ROCm
            backend = os.environ.get("KF_DIST_BACKEND", backend)  # test hook: 2 ranks sharing one GPU need gloo
            if torch.cuda.is_available() and not cpu:
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend=backend)
            self.num_processes, self.process_index = dist.get_world_size(), dist.get_rank()
            self.local_process_index = local_rank
        else:
            self.num_processes, self.process_index, self.local_process_index = 1, 0, 0
        if cpu or not torch.cuda.is_available():
            self.device = torch.device("cpu")
        else:
            index = self.local_process_index if self.num_processes > 1 else torch.cuda.current_device()
            self.device = torch.device("cuda", index % torch.cuda.device_count())
            torch.cuda.set_device(self.device)
        self.initialized = True

    def __repr__(self) -> str:
        return (f"State(num_processes={self.num_processes}, process_index={self.process_index}, "
                f"local_process_index={self.local_process_index}, device={self.device})")

    @classmethod
    def _reset_state(cls) -> None:
        cls._shared.clear()

    @property
    def use_distributed(self) -> bool:
        return self.num_processes > 1 or (force_exchanges() and dist.is_available() and dist.is_initialized())

    @property
    def is_main_process(self) -> bool:
        return self.process_index == 0

    @property
    def is_local_main_process(self) -> bool:
        return self.local_process_index == 0

    @property
    def is_last_process(self) -> bool:
        return self.process_index == self.num_processes - 1

    @property
    def default_device(self) -> torch.device:
        """The device a fresh ``State`` would compute on: the GPU when there is one."""
        return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def wait_for_everyone(self) -> None:
        if self.use_distributed:
            dist.barrier()


@contextlib.contextmanager
def paused_gc():
    """Suspends CPython's cyclic garbage collector for the duration of a stage loop.  The hooks
    allocate thousands of short-lived tensor wrappers per batch; an automatic generation-2
    collection in the middle of a stage stalls the launch thread for tens of milliseconds while the
    GPU idles (measured: every second MNIST-MLP pairwise step 20-60 ms slower).  Reference counting
    still frees tensors immediately; cycles are collected once, when the stage ends."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


# Score blocks leave the device through PINNED host memory (torch's caching host allocator: blocks are reused, never unmapped).
# ``tensor.cpu()`` allocates fresh pageable memory for every result: glibc serves such sizes by mmap / munmap, and on this stack an
# unmap of memory the GPU driver has pinned for a pageable copy stalls the process for tens of milliseconds -- every third
# MNIST-MLP pairwise step took 78 ms instead of 10.4 (a 65 ms "Memcpy DtoH" of 400 KB, an 84 ms ``torch.cat`` of one chunk:
# tools/r06_timeline.py) -- and a pageable copy goes through bounce buffers (200 MB of ResNet-9 scores per step).  Results above
# ``PINNED_HOST_LIMIT`` bytes stay pageable (pinned memory is not swappable); ``KF_PINNED_SCORES=0`` switches the old path back.
PINNED_HOST_LIMIT = 4 << 30


def to_host(tensor: torch.Tensor) -> torch.Tensor:
    """The tensor in host memory (as ``tensor.cpu()``), through a pinned buffer when it is a device tensor of moderate size."""
    if not tensor.is_cuda:
        return tensor.cpu()
    nbytes = tensor.numel() * tensor.element_size()
    if nbytes == 0 or nbytes > PINNED_HOST_LIMIT or os.environ.get("KF_PINNED_SCORES", "1") == "0":
        return tensor.cpu()
    out = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    out.copy_(tensor)   # blocking: the result is complete on return, as with .cpu()
    return out


def release_memory() -> None:
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


@contextlib.contextmanager
def no_sync(model: torch.nn.Module, state: Optional[State] = None):
    """Every parameter is frozen (``prepare_model``), so a DDP wrapper never has gradients to
    all-reduce; kept for API symmetry with the reference (utils/state.py:142-165)."""
    ctx = getattr(model, "no_sync", None)
    if callable(ctx):
        with ctx():
            yield
    else:
        yield
