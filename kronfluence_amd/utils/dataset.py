"""Dataset plumbing: loader kwargs, index partitions, and the two samplers that define how the
training set is sharded over ranks (reference ``utils/dataset.py:104-199``).

* ``DistributedEvalSampler``: strided shard ``rank, rank+P, ...`` with NO padding -- used for factor
  fitting so that counts are exact.
* ``DistributedSamplerWithStack``: contiguous chunk of ``ceil(N/P)`` per rank, wrap-around padded --
  used for the train side of pairwise scoring so that rank blocks concatenate in dataset order.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from collections.abc import Mapping
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch.utils import data


@dataclass
class DataLoaderKwargs:
    num_workers: int = 0
    collate_fn: Optional[Callable] = None
    pin_memory: bool = False
    timeout: int = 0
    worker_init_fn: Optional[Callable] = None
    multiprocessing_context: Optional[Any] = None
    generator: Optional[torch.Generator] = None
    prefetch_factor: Optional[int] = None
    persistent_workers: bool = False
    pin_memory_device: str = ""

    def to_dict(self) -> Dict[str, Any]:
        return dict(vars(self))


def partition_sizes(total: int, parts: int) -> List[int]:
    """Sizes of ``parts`` near-equal bins covering ``total`` items: the first ``total % parts`` bins hold one
    item more (what ``np.array_split`` / the divmod slices of the reference produce,
    ``utils/dataset.py:54-63`` and ``module/utils.py:125-131``) -- per-partition files are therefore
    interchangeable with the reference's."""
    base, extra = divmod(total, parts)
    return [base + (1 if i < extra else 0) for i in range(parts)]


def make_indices_partition(total_data_examples: int, partition_size: int) -> List[Tuple[int, int]]:
    """``[start, end)`` ranges of near-equal size (remainder spread over the leading ranges)."""
    if total_data_examples < partition_size:
        raise ValueError("The total data examples must be equal to or greater than the partition size.")
    bounds, start = [], 0
    for size in partition_sizes(total_data_examples, partition_size):
        bounds.append((start, start + size))
        start += size
    return bounds


def find_batch_size(batch: Any) -> int:
    """Leading dimension of the first tensor found in a (possibly nested) batch -- tensors, any ``Mapping``
    (HF ``BatchEncoding`` is a ``UserDict``), lists / tuples / namedtuples."""
    size = _find_batch_size(batch)
    if size is None:
        raise TypeError(f"Cannot find the batch size of a batch of type {type(batch).__name__}: no tensor inside.")
    return size


def _find_batch_size(batch: Any) -> Optional[int]:
    if isinstance(batch, torch.Tensor):
        return batch.shape[0] if batch.dim() > 0 else None
    if isinstance(batch, Mapping):
        batch = list(batch.values())
    if isinstance(batch, (list, tuple)):
        for value in batch:
            size = _find_batch_size(value)
            if size is not None:
                return size
    return None


def send_to_device(batch: Any, device: torch.device) -> Any:
    """Moves every tensor of a nested batch; containers keep their type (namedtuples included), any object with a
    ``.to(device)`` method (``BatchEncoding``) is asked to move itself."""
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, tuple) and hasattr(batch, "_fields"):  # namedtuple: positional constructor
        return type(batch)(*(send_to_device(v, device) for v in batch))
    if isinstance(batch, (list, tuple)):
        return type(batch)(send_to_device(v, device) for v in batch)
    if isinstance(batch, dict):
        return type(batch)({k: send_to_device(v, device) for k, v in batch.items()})
    if hasattr(batch, "to") and callable(batch.to):
        try:
            return batch.to(device)
        except TypeError:
            pass
    if isinstance(batch, Mapping):
        return {k: send_to_device(v, device) for k, v in batch.items()}
    return batch


def _resolve(num_replicas: Optional[int], rank: Optional[int]) -> Tuple[int, int]:
    if num_replicas is None or rank is None:
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("Requires an initialised torch.distributed process group.")
        num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
        rank = dist.get_rank() if rank is None else rank
    if not 0 <= rank < num_replicas:
        raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}].")
    return num_replicas, rank


class DistributedEvalSampler(data.Sampler):
    """Every ``P``-th example starting at ``rank``; shards are disjoint and cover the dataset once."""

    def __init__(self, dataset: data.Dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None,
                 seed: int = 0) -> None:
        self.num_replicas, self.rank = _resolve(num_replicas, rank)
        self.dataset, self.seed = dataset, seed
        self.total_size = len(dataset)
        self.num_samples = len(range(self.rank, self.total_size, self.num_replicas))

    def __iter__(self) -> Iterator[int]:
        return iter(range(self.rank, self.total_size, self.num_replicas))

    def __len__(self) -> int:
        return self.num_samples


class DistributedSamplerWithStack(data.Sampler):
    """Rank ``r`` gets the contiguous block ``[r*c, (r+1)*c)`` of the wrap-padded index list, ``c = ceil(N/P)``."""

    def __init__(self, dataset: data.Dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None,
                 seed: int = 0) -> None:
        self.num_replicas, self.rank = _resolve(num_replicas, rank)
        self.dataset, self.seed, self.epoch = dataset, seed, 0
        self.num_samples = math.ceil(len(dataset) / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self) -> Iterator[int]:
        n = len(self.dataset)
        start = self.rank * self.num_samples
        return iter([(start + j) % n for j in range(self.num_samples)])

    def __len__(self) -> int:
        return self.num_samples


class ResidentLoader:
    """Batches sliced zero-copy out of tensors already resident in HBM (288 GB per MI355X holds the
    whole dataset for every BASELINE config but the largest): no worker processes, no collation, no
    H2D copy per batch.  Quacks like a ``DataLoader`` as far as the stage loops are concerned
    (``__iter__``, ``__len__``, ``.dataset``, ``.sampler``).  ``indices`` (a sampler's index list)
    selects and orders the rows once, up front."""

    def __init__(self, tensors: Tuple[torch.Tensor, ...], batch_size: int, indices: Optional[List[int]] = None) -> None:
        self.dataset = data.TensorDataset(*tensors)
        if indices is not None:
            index = torch.as_tensor(list(indices), dtype=torch.int64, device=tensors[0].device)
            tensors = tuple(t.index_select(0, index) for t in tensors)
        self.tensors = tensors
        self.batch_size = batch_size
        self.sampler = range(tensors[0].shape[0])

    def __len__(self) -> int:
        return math.ceil(len(self.sampler) / self.batch_size)

    def __iter__(self):
        n = len(self.sampler)
        for start in range(0, n, self.batch_size):
            yield tuple(t[start:start + self.batch_size] for t in self.tensors)


def is_out_of_memory(exc: Exception) -> bool:
    """Whether ``exc`` is the device running out of HBM (HIP reports it as ``torch.cuda.OutOfMemoryError`` or as a RuntimeError
    naming hipErrorOutOfMemory)."""
    if isinstance(exc, torch.cuda.OutOfMemoryError):
        return True
    text = str(exc).lower()
    return isinstance(exc, RuntimeError) and ("out of memory" in text or "hiperroroutofmemory" in text)


def find_executable_batch_size(func: Callable[[int], object], start_batch_size: int) -> int:
    """Largest batch size, halving from ``start_batch_size``, for which ``func(batch_size)`` does not run out of device memory;
    any other exception propagates, reaching zero raises (reference ``utils/dataset.py:66-101``)."""
    batch_size = max(int(start_batch_size), 0)
    while True:
        if batch_size == 0:
            raise RuntimeError("No executable batch size found, reached zero.")
        try:
            func(batch_size)
        except Exception as exc:  # noqa: BLE001 -- only memory exhaustion is retried
            if not is_out_of_memory(exc):
                raise
            batch_size //= 2
            continue
        return batch_size
