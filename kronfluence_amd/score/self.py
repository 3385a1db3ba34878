"""Self-influence stage (reference ``score/self.py:135-443``) and its safetensors layout
(``:38-132``): ``score_n = <P(grad loss_n), grad loss_n>``, or with ``use_measurement_for_self_influence``
``<P(grad measurement_n), grad loss_n>``.

One ``[N_shard]`` fp32 vector per output key lives in HBM for the whole pass; every tracked layer adds its
contribution at the batch offset (``kf_rowwise_dot``), and the vector crosses PCIe once, at the end.
"""

from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional

import torch
from safetensors.torch import load_file, save_file
from torch import autocast, nn
from torch.utils import data

from kronfluence_amd.arguments import FactorArguments, ScoreArguments
from kronfluence_amd.factor.covariance import _loss_scale
from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
from kronfluence_amd.module.utils import (
    READ_ONLY_FACTORS_WHEN_SCORING,
    finalize_all_iterations, finalize_iteration, get_tracked_module_names, prepare_modules, set_factors,
    set_gradient_scale, set_mode, set_score_sink, update_factor_args, update_score_args,
)
from kronfluence_amd.score.dot_product import gather_score_blocks
from kronfluence_amd.task import Task
from kronfluence_amd.utils.constants import ALL_MODULE_NAME, FACTOR_TYPE, SCORE_TYPE
from kronfluence_amd.utils.dataset import find_batch_size, send_to_device
from kronfluence_amd.utils.state import State, no_sync, paused_gc


def self_scores_save_path(output_dir: Path, partition=None) -> Path:
    if partition is not None:
        return output_dir / f"self_scores_data_partition{partition[0]}_module_partition{partition[1]}.safetensors"
    return output_dir / "self_scores.safetensors"


def save_self_scores(output_dir: Path, scores: SCORE_TYPE, partition=None, metadata: Optional[Dict[str, str]] = None) -> None:
    save_file(tensors={k: v.contiguous() for k, v in scores.items()},
              filename=str(self_scores_save_path(output_dir, partition)), metadata=metadata)


def load_self_scores(output_dir: Path, partition=None) -> SCORE_TYPE:
    return load_file(filename=str(self_scores_save_path(output_dir, partition)))


def self_scores_exist(output_dir: Path, partition=None) -> bool:
    return self_scores_save_path(output_dir, partition).exists()


def _self_scores_impl(loaded_factors: FACTOR_TYPE, model: nn.Module, state: State, task: Task,
                      train_loader: data.DataLoader, score_args: ScoreArguments, factor_args: FactorArguments,
                      tracked_module_names: Optional[List[str]], with_measurement: bool) -> SCORE_TYPE:
    update_factor_args(model, factor_args)
    update_score_args(model, score_args)
    if tracked_module_names is None:
        tracked_module_names = get_tracked_module_names(model)
    score_mode = ModuleMode.SELF_MEASUREMENT_SCORE if with_measurement else ModuleMode.SELF_SCORE
    set_mode(model, score_mode, tracked_module_names, release_memory=True)
    for name in loaded_factors:
        set_factors(model, name, loaded_factors[name], clone=True, share=READ_ONLY_FACTORS_WHEN_SCORING)
    prepare_modules(model, tracked_module_names, state.device)

    modules = [m for m in model.modules() if isinstance(m, TrackedModule) and m.name in tracked_module_names]
    shard_size = len(train_loader.sampler) if hasattr(train_loader, "sampler") else len(train_loader.dataset)
    dataset_size = len(train_loader.dataset)
    keys = [m.name for m in modules] if score_args.compute_per_module_scores else [ALL_MODULE_NAME]
    buffers = {key: torch.zeros(shard_size, dtype=torch.float32, device=state.device) for key in keys}

    enable_amp = score_args.amp_dtype is not None
    scale = _loss_scale(factor_args) if (enable_amp and factor_args.amp_dtype == torch.float16) else 1.0
    if scale != 1.0:
        set_gradient_scale(model, 1.0 / scale)

    def backward(value: torch.Tensor) -> None:
        (value * scale if scale != 1.0 else value).backward()

    offset = 0
    for batch in train_loader:
        batch = send_to_device(batch, state.device)
        if with_measurement:
            # first backward: P(grad measurement) per sample, held in storage["preconditioned_gradient"]
            set_mode(model, ModuleMode.PRECONDITION_GRADIENT, tracked_module_names, release_memory=False)
            with no_sync(model, state):
                model.zero_grad(set_to_none=True)
                with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                    measurement = task.compute_measurement(batch=batch, model=model)
                backward(measurement)
            if factor_args.has_shared_parameters:
                finalize_iteration(model, tracked_module_names)
            del measurement
            set_mode(model, ModuleMode.SELF_MEASUREMENT_SCORE, tracked_module_names, release_memory=False)
        for m in modules:
            m.score_sink = (buffers[m.name if score_args.compute_per_module_scores else ALL_MODULE_NAME], offset)
        with no_sync(model, state):
            model.zero_grad(set_to_none=True)
            with autocast(device_type=state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                loss = task.compute_train_loss(batch=batch, model=model, sample=False)
            backward(loss)
        if factor_args.has_shared_parameters or with_measurement:
            finalize_iteration(model, tracked_module_names)
        offset += find_batch_size(batch)
        del loss

    model.zero_grad(set_to_none=True)
    set_score_sink(model, None, tracked_module_names)
    finalize_all_iterations(model, tracked_module_names)
    set_gradient_scale(model, 1.0)
    set_mode(model, ModuleMode.DEFAULT, tracked_module_names, release_memory=True)

    total: SCORE_TYPE = {}
    for key, vector in buffers.items():
        gathered = gather_score_blocks(vector.to(score_args.score_dtype).unsqueeze(0), state, dataset_size)
        total[key] = gathered.squeeze(0)
    state.wait_for_everyone()
    return total


def compute_self_scores_with_loaders(loaded_factors: FACTOR_TYPE, model: nn.Module, state: State, task: Task,
                                     train_loader: data.DataLoader, score_args: ScoreArguments,
                                     factor_args: FactorArguments, tracked_module_names: Optional[List[str]] = None,
                                     disable_tqdm: bool = False) -> SCORE_TYPE:
    """``{"all_modules": [N]}`` (or one vector per module) on rank 0 (reference ``score/self.py:135-290``)."""
    del disable_tqdm
    with paused_gc():
        return _self_scores_impl(loaded_factors, model, state, task, train_loader, score_args, factor_args,
                                 tracked_module_names, with_measurement=False)


def compute_self_measurement_scores_with_loaders(loaded_factors: FACTOR_TYPE, model: nn.Module, state: State, task: Task,
                                                 train_loader: data.DataLoader, score_args: ScoreArguments,
                                                 factor_args: FactorArguments,
                                                 tracked_module_names: Optional[List[str]] = None,
                                                 disable_tqdm: bool = False) -> SCORE_TYPE:
    """Measurement variant (reference ``score/self.py:293-443``): two backward passes per batch."""
    del disable_tqdm
    with paused_gc():
        return _self_scores_impl(loaded_factors, model, state, task, train_loader, score_args, factor_args,
                                 tracked_module_names, with_measurement=True)
