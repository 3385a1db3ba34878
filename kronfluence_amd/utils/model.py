"""Process-group helpers with the reference's names (``kronfluence/utils/model.py``): scripts written for the reference call
``apply_ddp(model, local_rank, rank, world_size)`` after ``prepare_model`` and hand the result to the ``Analyzer``.

On this engine the DATA is sharded and the model replicated: every parameter is frozen by ``prepare_model``, so the wrapper never
has a gradient to all-reduce (``utils/state.py:no_sync`` is kept for symmetry) -- it is a replica container whose constructor
broadcasts rank 0's parameters and buffers, exactly what the reference uses it for.  ``"nccl"`` is RCCL on ROCm,
one process per MI355X.  FSDP (a memory trick of the reference for models that do not fit one device) is outside the hot path this
engine accelerates -- 288 GB of HBM per GPU hold the configurations it targets -- and is refused by name, not silently ignored.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.parallel import DistributedDataParallel

from kronfluence_amd.utils.state import State


def apply_ddp(model: nn.Module, local_rank: int, rank: int, world_size: int) -> DistributedDataParallel:
    """Initialises the process group (RCCL; ``KF_DIST_BACKEND`` or a GPU-less machine selects gloo), pins this process to GPU
    ``local_rank``, moves ``model`` there and returns it wrapped.  Reference: utils/model.py:17-55."""
    on_gpu = torch.cuda.is_available()
    if not dist.is_initialized():
        backend = os.environ.get("KF_DIST_BACKEND", "nccl" if on_gpu else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    State._reset_state()   # a State created before the group existed would still say "one process"
    if on_gpu:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        model = model.to(device=device)
        return DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank)
    return DistributedDataParallel(model)


def apply_fsdp(*args, **kwargs):
    """Not provided: see the module docstring (SURVEY.md row 14 / C11 mark FSDP out of scope)."""
    raise NotImplementedError(
        "apply_fsdp is not part of kronfluence_amd: the engine shards the data and replicates the model (one process per MI355X, "
        "288 GB of HBM each); use apply_ddp, or no wrapper at all under torchrun.")
