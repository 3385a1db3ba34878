"""Stage 3 outer loop: preconditioned query gradients, then train passes (reference
``score/pairwise.py:133-293``), plus the scores' safetensors layout (``:38-130``)."""

from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional

import torch
from safetensors.torch import load_file, save_file
from torch import autocast, nn
from torch.utils import data

from kronfluence_amd.arguments import FactorArguments, ScoreArguments, unsupported_score_options
from kronfluence_amd.factor.covariance import _loss_scale
from kronfluence_amd.module.tracked_module import ModuleMode
from kronfluence_amd.module.utils import (
    READ_ONLY_FACTORS_WHEN_SCORING,
    accumulate_iterations, finalize_all_iterations, finalize_iteration, get_tracked_module_names, prepare_modules,
    set_async_query_gather, set_factors, set_gradient_scale, set_mode, set_query_capacity, synchronize_modules, truncate,
    update_factor_args,
    update_score_args,
)
from kronfluence_amd.score.query_exchange import is_replicated
from kronfluence_amd.score.dot_product import (
    compute_aggregated_dot_products_with_loader, compute_dot_products_with_loader,
)
from kronfluence_amd.task import Task
from kronfluence_amd.utils.constants import FACTOR_TYPE, SCORE_TYPE
from kronfluence_amd.utils.dataset import send_to_device
from kronfluence_amd.utils.state import State, no_sync, paused_gc


# Measurement hook (bench.py): when a dict, the stage adds the wall seconds of its two phases -- query phase (measurement passes +
# preconditioner + exchange) and train passes -- to ``query_s`` / ``train_s``, with ONE device synchronisation per held-query window.
STAGE_LOG: Optional[Dict[str, float]] = None


def _mark(key: Optional[str], since: float) -> float:
    import time

    if STAGE_LOG is None:
        return since
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    now = time.perf_counter()
    if key is not None:
        STAGE_LOG[key] = STAGE_LOG.get(key, 0.0) + (now - since)
    return now


def pairwise_scores_save_path(output_dir: Path, partition=None) -> Path:
    if partition is not None:
        return output_dir / f"pairwise_scores_data_partition{partition[0]}_module_partition{partition[1]}.safetensors"
    return output_dir / "pairwise_scores.safetensors"


def save_pairwise_scores(output_dir: Path, scores: SCORE_TYPE, partition=None, metadata: Optional[Dict[str, str]] = None) -> None:
    save_file(tensors={k: v.contiguous() for k, v in scores.items()},
              filename=str(pairwise_scores_save_path(output_dir, partition)), metadata=metadata)


def load_pairwise_scores(output_dir: Path, partition=None) -> SCORE_TYPE:
    return load_file(filename=str(pairwise_scores_save_path(output_dir, partition)))


def pairwise_scores_exist(output_dir: Path, partition=None) -> bool:
    return pairwise_scores_save_path(output_dir, partition).exists()


def _compute_pairwise_scores_with_loaders_impl(loaded_factors: FACTOR_TYPE, model: nn.Module, state: State, task: Task,
                                         query_loader: data.DataLoader, per_device_query_batch_size: int,
                                         train_loader: data.DataLoader, score_args: ScoreArguments,
                                         factor_args: FactorArguments, tracked_module_names: Optional[List[str]],
                                         disable_tqdm: bool = False) -> SCORE_TYPE:
    flagged = unsupported_score_options(score_args)
    if flagged:
        raise NotImplementedError(
            f"ScoreArguments options {flagged} are outside the MI355X pairwise hot path (SURVEY.md section 8f)."
        )
    update_factor_args(model, factor_args)
    update_score_args(model, score_args)
    if tracked_module_names is None:
        tracked_module_names = get_tracked_module_names(model)
    set_mode(model, ModuleMode.PRECONDITION_GRADIENT, tracked_module_names, release_memory=True)
    for name in loaded_factors:
        set_factors(model, name, loaded_factors[name], clone=True, share=READ_ONLY_FACTORS_WHEN_SCORING)
    prepare_modules(model, tracked_module_names, state.device)

    chunks: Dict[str, List[torch.Tensor]] = {}
    # Multi-rank query side (score/query_exchange.py): ``gather`` -- the reference's strided query shard + per-layer all-gather
    # (precondition.py:166-214) -- or ``replicate`` -- the loader hands EVERY rank all queries in dataset order and nothing is
    # exchanged.  Either way a window holds ``accumulation_steps x P x q`` queries per train pass, so the score chunks are the same.
    replicated = bool(state.use_distributed and is_replicated(query_loader))
    total_query_batch_size = per_device_query_batch_size * (1 if replicated else state.num_processes)
    steps_per_window = score_args.query_gradient_accumulation_steps * (state.num_processes if replicated else 1)
    query_remainder = len(query_loader.dataset) % total_query_batch_size
    num_batches = len(query_loader)
    expected = -(-len(query_loader.dataset) // total_query_batch_size)
    if state.use_distributed and num_batches != expected:
        raise ValueError(f"The query loader yields {num_batches} batches where {expected} are expected of a "
                         f"{'replicated (unsharded)' if replicated else 'rank-sharded'} loader: build it with the sampler that matches "
                         "`query_exchange.mark_replicated`.")
    enable_amp = score_args.amp_dtype is not None
    scale = _loss_scale(factor_args) if (enable_amp and factor_args.amp_dtype == torch.float16) else 1.0
    if scale != 1.0:
        set_gradient_scale(model, 1.0 / scale)

    held = 0
    remaining = len(query_loader.dataset)  # queries still to be preconditioned (each rank ends up holding all of them)
    window = steps_per_window * total_query_batch_size
    set_query_capacity(model, tracked_module_names, min(window, remaining))
    set_async_query_gather(model, tracked_module_names, bool(state.use_distributed and not replicated))
    mark = _mark(None, 0.0)
    try:
        for query_index, query_batch in enumerate(query_loader):
            query_batch = send_to_device(query_batch, state.device)
            with no_sync(model, state):
                model.zero_grad(set_to_none=True)
                with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                    measurement = task.compute_measurement(batch=query_batch, model=model)
                (measurement * scale if scale != 1.0 else measurement).backward()
            if factor_args.has_shared_parameters:
                finalize_iteration(model, tracked_module_names)
            if state.use_distributed and not replicated:
                synchronize_modules(model, tracked_module_names, num_processes=state.num_processes)  # C4
                if query_index == num_batches - 1 and query_remainder > 0:
                    truncate(model, tracked_module_names, keep_size=query_remainder)
            accumulate_iterations(model, tracked_module_names)
            del query_batch, measurement
            held += 1
            if held < steps_per_window and query_index != num_batches - 1:
                continue
            dot_products = (compute_aggregated_dot_products_with_loader if score_args.aggregate_train_gradients
                            else compute_dot_products_with_loader)
            mark = _mark("query_s", mark)
            scores = dot_products(model=model, state=state, task=task, train_loader=train_loader, factor_args=factor_args,
                                  score_args=score_args, tracked_module_names=tracked_module_names, loss_scale=scale)
            mark = _mark("train_s", mark)
            if state.is_main_process:
                for key, value in scores.items():
                    chunks.setdefault(key, []).append(value)
            del scores
            state.wait_for_everyone()
            held = 0
            remaining -= window
            set_query_capacity(model, tracked_module_names, min(window, max(remaining, 0)))
    finally:
        set_query_capacity(model, tracked_module_names, None)  # also when the score loop raises
        set_async_query_gather(model, tracked_module_names, False)

    total: SCORE_TYPE = {}
    if state.is_main_process:
        # (one chunk -- all queries held at once -- is handed over as it is: no second host copy)
        total = {key: parts[0] if len(parts) == 1 else torch.cat(parts, dim=0) for key, parts in chunks.items()}
    model.zero_grad(set_to_none=True)
    set_gradient_scale(model, 1.0)
    set_query_capacity(model, tracked_module_names, None)
    finalize_all_iterations(model, tracked_module_names)
    set_mode(model, ModuleMode.DEFAULT, release_memory=True)
    state.wait_for_everyone()
    return total


def _compute_pairwise_query_aggregated_scores_impl(loaded_factors: FACTOR_TYPE, model: nn.Module, state: State, task: Task,
                                                   query_loader: data.DataLoader, per_device_query_batch_size: int,
                                                   train_loader: data.DataLoader, score_args: ScoreArguments,
                                                   factor_args: FactorArguments, tracked_module_names: Optional[List[str]],
                                                   disable_tqdm: bool = False) -> SCORE_TYPE:
    """``aggregate_query_gradients`` (reference ``score/pairwise.py:296-393``): the query pass sums the raw query
    gradients (``GradientTracker``), the sum is preconditioned once, and a single train pass scores against it."""
    del per_device_query_batch_size, disable_tqdm
    flagged = unsupported_score_options(score_args)
    if flagged:
        raise NotImplementedError(
            f"ScoreArguments options {flagged} are outside the MI355X pairwise hot path (SURVEY.md section 8f)."
        )
    update_factor_args(model, factor_args)
    update_score_args(model, score_args)
    if tracked_module_names is None:
        tracked_module_names = get_tracked_module_names(model)
    set_mode(model, ModuleMode.GRADIENT_AGGREGATION, tracked_module_names, release_memory=True)
    for name in loaded_factors:
        set_factors(model, name, loaded_factors[name], clone=True, share=READ_ONLY_FACTORS_WHEN_SCORING)
    prepare_modules(model, tracked_module_names, state.device)
    enable_amp = score_args.amp_dtype is not None
    scale = _loss_scale(factor_args) if (enable_amp and factor_args.amp_dtype == torch.float16) else 1.0
    if scale != 1.0:
        set_gradient_scale(model, 1.0 / scale)
    for query_batch in query_loader:
        query_batch = send_to_device(query_batch, state.device)
        with no_sync(model, state):
            model.zero_grad(set_to_none=True)
            with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                measurement = task.compute_measurement(batch=query_batch, model=model)
            (measurement * scale if scale != 1.0 else measurement).backward()
        if factor_args.has_shared_parameters:
            finalize_iteration(model, tracked_module_names)
        del query_batch, measurement
    if state.use_distributed:
        synchronize_modules(model, tracked_module_names, num_processes=state.num_processes)
    set_mode(model, ModuleMode.PRECONDITION_GRADIENT, tracked_module_names, release_memory=False)
    finalize_all_iterations(model, tracked_module_names)  # precondition the summed gradient -> one held query row
    dot_products = (compute_aggregated_dot_products_with_loader if score_args.aggregate_train_gradients
                    else compute_dot_products_with_loader)
    scores = dot_products(model=model, state=state, task=task, train_loader=train_loader, factor_args=factor_args,
                          score_args=score_args, tracked_module_names=tracked_module_names, loss_scale=scale)
    model.zero_grad(set_to_none=True)
    set_gradient_scale(model, 1.0)
    set_mode(model, ModuleMode.DEFAULT, release_memory=True)
    state.wait_for_everyone()
    return scores if state.is_main_process else {}


def compute_pairwise_query_aggregated_scores_with_loaders(*args, **kwargs) -> SCORE_TYPE:
    """Stage entry point of the query-aggregated variant, cyclic GC paused."""
    with paused_gc():
        return _compute_pairwise_query_aggregated_scores_impl(*args, **kwargs)


def compute_pairwise_scores_with_loaders(*args, **kwargs) -> SCORE_TYPE:
    """Stage entry point (signature of ``_compute_pairwise_scores_with_loaders_impl``), cyclic GC paused."""
    with paused_gc():
        return _compute_pairwise_scores_with_loaders_impl(*args, **kwargs)
