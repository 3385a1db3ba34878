"""Model-wide helpers that fan an operation out to every ``TrackedModule`` (reference
``module/utils.py:33-413``), plus the bucketed RCCL exchange of the factor stage."""

from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.parallel import DataParallel, DistributedDataParallel

from kronfluence_amd.arguments import FactorArguments, ScoreArguments
from kronfluence_amd.module.conv2d import TrackedConv2d  # noqa: F401  (registers nn.Conv2d)
from kronfluence_amd.module.linear import TrackedLinear  # noqa: F401  (registers nn.Linear)
from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
from kronfluence_amd.task import Task
from kronfluence_amd.utils.comm import exchange
from kronfluence_amd.utils.constants import FACTOR_TYPE
from kronfluence_amd.utils.exceptions import IllegalTaskConfigurationError, TrackedModuleNotFoundError


def _tracked(model: nn.Module, names: Optional[Iterable[str]] = None) -> List[TrackedModule]:
    wanted = None if names is None else set(names)
    return [m for m in model.modules() if isinstance(m, TrackedModule) and (wanted is None or m.name in wanted)]


def wrap_tracked_modules(model: nn.Module, task: Optional[Task] = None, factor_args: Optional[FactorArguments] = None,
                         score_args: Optional[ScoreArguments] = None) -> nn.Module:
    """Replaces every supported leaf (optionally only those the task names) by its ``TrackedModule``."""
    if isinstance(model, (DataParallel, DistributedDataParallel)) or type(model).__name__ == "FullyShardedDataParallel":
        raise ValueError(
            "The model is wrapped with DataParallel, DistributedDataParallel or FullyShardedDataParallel. "
            "Call `prepare_model` before wrapping the model."
        )
    requested = task.get_influence_tracked_modules() if task is not None else None
    found = {name: False for name in requested} if requested is not None else None
    process_fnc = task.post_process_per_sample_gradient if (task is not None and task.enable_post_process_per_sample_gradient) else None
    supported = tuple(TrackedModule.SUPPORTED_MODULES)
    for name, module in list(model.named_modules()):
        if any(True for _ in module.children()):
            continue
        if requested is not None and name not in found:
            continue
        if isinstance(module, supported):
            wrapper = TrackedModule.SUPPORTED_MODULES[type(module)](
                name=name, original_module=module, factor_args=factor_args, score_args=score_args,
                per_sample_gradient_process_fnc=process_fnc)
            parent = model.get_submodule(".".join(name.split(".")[:-1])) if "." in name else model
            setattr(parent, name.split(".")[-1], wrapper)
            if found is not None:
                found[name] = True
    if found is not None and not all(found.values()):
        raise IllegalTaskConfigurationError(f"Some provided tracked modules were not found. The current mapping: `{found}`.")
    if not _tracked(model):
        kinds = ", ".join(t.__name__ for t in TrackedModule.SUPPORTED_MODULES)
        raise IllegalTaskConfigurationError(
            f"No supported modules found. Supported module types: {kinds}. Consider rewriting your model or "
            f"subclassing `TrackedModule` for custom layers.\nCurrent Model:\n{model}"
        )
    return model


def get_tracked_module_names(model: nn.Module) -> List[str]:
    return [m.name for m in _tracked(model)]


def make_modules_partition(total_module_names: List[str], partition_size: int) -> List[List[str]]:
    """Near-equal consecutive groups, remainder spread over the leading groups (reference ``module/utils.py:125-131``)."""
    if len(total_module_names) < partition_size:
        raise ValueError("The total modules must be equal to or greater than the partition size.")
    from kronfluence_amd.utils.dataset import partition_sizes

    groups, start = [], 0
    for size in partition_sizes(len(total_module_names), partition_size):
        groups.append(total_module_names[start:start + size])
        start += size
    return groups


def update_factor_args(model: nn.Module, factor_args: FactorArguments) -> None:
    for m in _tracked(model):
        m.update_factor_args(factor_args)


def update_score_args(model: nn.Module, score_args: ScoreArguments) -> None:
    for m in _tracked(model):
        m.update_score_args(score_args)


def set_mode(model: nn.Module, mode: str, tracked_module_names: Optional[List[str]] = None,
             release_memory: bool = False) -> None:
    for m in _tracked(model, tracked_module_names):
        m.set_mode(mode=mode, release_memory=release_memory)


def set_attention_mask(model: nn.Module, attention_mask: Optional[Any] = None) -> None:
    for m in _tracked(model):
        if isinstance(attention_mask, dict):
            m.set_attention_mask(attention_mask.get(m.name))
        else:
            m.set_attention_mask(attention_mask)


def set_gradient_scale(model: nn.Module, gradient_scale: float = 1.0) -> None:
    for m in _tracked(model):
        m.set_gradient_scale(gradient_scale)


def set_score_sink(model: nn.Module, sink, tracked_module_names: Optional[List[str]] = None) -> None:
    for m in _tracked(model, tracked_module_names):
        m.score_sink = sink


def prepare_modules(model: nn.Module, tracked_module_names: List[str], device: torch.device) -> None:
    for m in _tracked(model, tracked_module_names):
        m.prepare_storage(device=device)


def load_factors(model: nn.Module, factor_name: str, tracked_module_names: Optional[List[str]] = None,
                 cpu: bool = True, dtype: Optional[torch.dtype] = None) -> Dict[str, torch.Tensor]:
    """``{module_name: factor}`` for every module holding ``factor_name`` (reference ``utils.py:201-235``).
    ``dtype`` casts floating factors on export (accumulators are fp32 on the device)."""
    out = {}
    for m in _tracked(model, tracked_module_names):
        factor = m.get_factor(factor_name)
        if factor is None:
            continue
        if dtype is not None and factor.is_floating_point() and factor.dtype != dtype:
            factor = factor.to(dtype)
        out[m.name] = factor.to("cpu") if cpu else factor
    return out


# Factors a stage only ever READS once they are set -- the rotations read the eigenvector matrices, ``FactorConfig.prepare`` replaces
# the Lambda matrix by a new tensor (1 / (Lambda / n + damping)) -- are SHARED with the caller's dictionary when they already live on
# the accelerator instead of cloned (reference module/utils.py:158-177 clones everything): at Llama-3-8B's full depth the clones are
# 99 GB of eigenvectors + 28 GB of Lambda on top of the caller's own 127 GB, the difference between fitting one MI355X and not.
# Everything else (counters, accumulators a stage adds to in place) is cloned as before.
READ_ONLY_FACTORS = ("activation_eigenvectors", "gradient_eigenvectors")
READ_ONLY_FACTORS_WHEN_SCORING = READ_ONLY_FACTORS + ("lambda_matrix",)


def set_factors(model: nn.Module, factor_name: str, factors: Dict[str, torch.Tensor], clone: bool = False,
                share: Iterable[str] = ()) -> None:
    """``share``: factor names whose accelerator-resident tensors are handed over as they are even with ``clone``."""
    shared = factor_name in tuple(share)
    for m in _tracked(model):
        if m.name in factors:
            factor = factors[m.name]
            keep = not clone or (shared and factor.is_cuda)
            m.set_factor(factor_name, factor if keep else factor.clone())


def factors_exist(model: nn.Module, tracked_module_names: Optional[List[str]] = None) -> bool:
    return all(m.exist() for m in _tracked(model, tracked_module_names))


def synchronize_modules(model: nn.Module, tracked_module_names: List[str], num_processes: int = 1) -> None:
    for m in _tracked(model, tracked_module_names):
        m.synchronize(num_processes=num_processes)


FACTOR_BUCKET_BYTES = 1 << 30  # per all-reduce: large enough to run at xGMI link speed, small enough to stage in HBM


def synchronize_factors(model: nn.Module, factor_names: List[str], tracked_module_names: List[str],
                        device: torch.device, extra: Optional[List[torch.Tensor]] = None,
                        bucket_bytes: Optional[int] = None) -> None:
    """C1-C3 as a handful of large exchanges: the floating factors of all layers are packed, in order, into flat fp32
    buckets of at most ``bucket_bytes`` (default 1 GiB; a factor larger than that is reduced in place, on its own),
    every int64 counter (plus ``extra``) into one int64 bucket, and each bucket is summed with a single all-reduce
    (RCCL over xGMI on GPU, gloo in the CPU tests).  The reference issues 4 (resp. 2) ``dist.reduce`` calls per layer
    (``tracker/factor.py:136-142, 315-321``); one flat buffer for everything would double the footprint of
    Llama-scale covariances (85 GB)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    limit = FACTOR_BUCKET_BYTES if bucket_bytes is None else bucket_bytes
    floats, ints = [], list(extra or [])
    for m in _tracked(model, tracked_module_names):
        for name in factor_names:
            t = m.get_factor(name)
            if t is None:
                continue
            (floats if t.is_floating_point() else ints).append((m, name, t))

    def reduce_bucket(group, dtype) -> None:
        tensors = [item[2] if isinstance(item, tuple) else item for item in group]
        if not tensors:
            return
        flat = torch.cat([t.reshape(-1).to(device=device, dtype=dtype) for t in tensors])
        with exchange("factor_all_reduce", flat.numel() * flat.element_size()):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        offset = 0
        for item, t in zip(group, tensors):
            n = t.numel()
            reduced = flat[offset:offset + n].reshape(t.shape)
            offset += n
            if isinstance(item, tuple):
                owner, name, _ = item
                owner.set_factor(name, reduced.to(device=t.device, dtype=t.dtype).clone())
            else:
                t.copy_(reduced.to(device=t.device, dtype=t.dtype))

    bucket, held = [], 0
    for item in floats:
        t = item[2]
        size = t.numel() * 4
        if size > limit and t.dtype == torch.float32 and t.device == device and t.is_contiguous():
            with exchange("factor_all_reduce", size):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)  # big enough on its own: no staging copy
            continue
        if bucket and held + size > limit:
            reduce_bucket(bucket, torch.float32)
            bucket, held = [], 0
        bucket.append(item)
        held += size
    reduce_bucket(bucket, torch.float32)
    reduce_bucket(ints, torch.int64)


def set_query_capacity(model: nn.Module, tracked_module_names: Optional[List[str]], capacity: Optional[int]) -> None:
    """Announces how many preconditioned query gradients the modules are about to accumulate (``QueryBuffer``)."""
    for m in _tracked(model, tracked_module_names):
        m.query_capacity = capacity


def set_async_query_gather(model: nn.Module, tracked_module_names: Optional[List[str]], enabled: bool) -> None:
    """Lets (or stops letting) the PreconditionTrackers issue their query all-gather from the backward hook; set by the
    pairwise query loop only, around the passes it follows with ``synchronize_modules``."""
    for m in _tracked(model, tracked_module_names):
        m.async_query_gather = enabled


def set_side_stream(model: nn.Module, tracked_module_names: Optional[List[str]], enabled: bool) -> None:
    """Stage loops only: lets the trackers run their hooks' kernels beside the model's pass (``BaseTracker._run_beside``);
    switching it off joins the side stream first, so whatever the hooks accumulated is complete for the caller's stream."""
    for m in _tracked(model, tracked_module_names):
        if not enabled:
            for tracker in m._trackers.values():
                tracker._join_side()
        m.side_stream_ok = enabled


def truncate(model: nn.Module, tracked_module_names: List[str], keep_size: int) -> None:
    for m in _tracked(model, tracked_module_names):
        m.truncate(keep_size=keep_size)


def accumulate_iterations(model: nn.Module, tracked_module_names: List[str]) -> None:
    for m in _tracked(model, tracked_module_names):
        m.accumulate_iterations()


def finalize_iteration(model: nn.Module, tracked_module_names: List[str]) -> None:
    for m in _tracked(model, tracked_module_names):
        m.finalize_iteration()


def finalize_all_iterations(model: nn.Module, tracked_module_names: List[str]) -> None:
    for m in _tracked(model, tracked_module_names):
        m.finalize_all_iterations()


def exist_for_all_modules(model: nn.Module, tracked_module_names: List[str]) -> bool:
    return factors_exist(model, tracked_module_names)
