"""Stage 1 loop: activation / pseudo-gradient covariance fitting (reference
``factor/covariance.py:153-266``) and the safetensors layout of its results (``:35-96``)."""

from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
from safetensors.torch import load_file, save_file
from torch import autocast, nn
from torch.utils import data

from kronfluence_amd.arguments import FactorArguments
from kronfluence_amd.module.tracked_module import ModuleMode
from kronfluence_amd.module.utils import (
    get_tracked_module_names, load_factors, set_attention_mask, set_gradient_scale, set_mode, set_side_stream, synchronize_factors,
    update_factor_args,
)
from kronfluence_amd.task import Task
from kronfluence_amd.utils.constants import (
    ACTIVATION_COVARIANCE_MATRIX_NAME, COVARIANCE_FACTOR_NAMES, FACTOR_TYPE, GRADIENT_COVARIANCE_MATRIX_NAME,
)
from kronfluence_amd.utils.dataset import find_batch_size, send_to_device
from kronfluence_amd.utils.state import State, no_sync, paused_gc


def covariance_matrices_save_path(output_dir: Path, factor_name: str, partition=None) -> Path:
    assert factor_name in COVARIANCE_FACTOR_NAMES
    if partition is not None:
        return output_dir / f"{factor_name}_data_partition{partition[0]}_module_partition{partition[1]}.safetensors"
    return output_dir / f"{factor_name}.safetensors"


def save_covariance_matrices(output_dir: Path, factors: FACTOR_TYPE, partition=None, metadata: Optional[Dict[str, str]] = None) -> None:
    assert set(factors.keys()) == set(COVARIANCE_FACTOR_NAMES)
    for name in factors:
        save_file(tensors={k: v.contiguous() for k, v in factors[name].items()},
                  filename=str(covariance_matrices_save_path(output_dir, name, partition)), metadata=metadata)


def load_covariance_matrices(output_dir: Path, partition=None) -> FACTOR_TYPE:
    return {name: load_file(filename=str(covariance_matrices_save_path(output_dir, name, partition)))
            for name in COVARIANCE_FACTOR_NAMES}


def covariance_matrices_exist(output_dir: Path, partition=None) -> bool:
    return all(covariance_matrices_save_path(output_dir, name, partition).exists() for name in COVARIANCE_FACTOR_NAMES)


def _loss_scale(factor_args: FactorArguments) -> float:
    """fp16 AMP only: a fixed loss scale instead of a ``GradScaler`` object (no step ever happens);
    the hooks un-scale through ``gradient_scale`` exactly as the reference (covariance.py:196-202)."""
    if factor_args.amp_dtype == torch.float16:
        return float(factor_args.amp_scale)
    return 1.0


def _fit_covariance_matrices_with_loader_impl(model: nn.Module, state: State, task: Task, loader: data.DataLoader,
                                        factor_args: FactorArguments, tracked_module_names: Optional[List[str]] = None,
                                        disable_tqdm: bool = False, all_ranks: bool = False,
                                        cpu: bool = True) -> Tuple[torch.Tensor, FACTOR_TYPE]:
    """``all_ranks`` / ``cpu`` are this engine's additions: the covariances are ALL-reduced (not reduced to rank 0), so
    every rank already holds the sums -- ``all_ranks=True`` returns them everywhere (the reference hands factors to the
    other ranks through the file system) and ``cpu=False`` leaves them in HBM for the next stage."""
    del disable_tqdm
    update_factor_args(model, factor_args)
    if tracked_module_names is None:
        tracked_module_names = get_tracked_module_names(model)
    set_mode(model, ModuleMode.COVARIANCE, tracked_module_names, release_memory=True)
    num_data_processed = torch.zeros((1,), dtype=torch.int64)
    enable_amp = factor_args.amp_dtype is not None
    scale = _loss_scale(factor_args)
    if scale != 1.0:
        set_gradient_scale(model, 1.0 / scale)
    set_side_stream(model, tracked_module_names, True)   # the hooks' kernels may run beside the model's passes ...
    try:
        for batch in loader:
            batch = send_to_device(batch, state.device)
            attention_mask = task.get_attention_mask(batch=batch)
            if attention_mask is not None:
                set_attention_mask(model, attention_mask)
            with no_sync(model, state):
                model.zero_grad(set_to_none=True)
                with autocast(device_type=state.device.type, enabled=enable_amp, dtype=factor_args.amp_dtype):
                    loss = task.compute_train_loss(batch=batch, model=model, sample=not factor_args.use_empirical_fisher)
                (loss * scale if scale != 1.0 else loss).backward()
            num_data_processed.add_(find_batch_size(batch))
            del loss
    finally:
        set_side_stream(model, tracked_module_names, False)   # ... and are joined before the factors are read
    if state.use_distributed:
        synchronize_factors(model, COVARIANCE_FACTOR_NAMES, tracked_module_names, state.device, extra=[num_data_processed])
    saved: FACTOR_TYPE = {}
    if state.is_main_process or all_ranks:
        dtypes = {ACTIVATION_COVARIANCE_MATRIX_NAME: factor_args.activation_covariance_dtype,
                  GRADIENT_COVARIANCE_MATRIX_NAME: factor_args.gradient_covariance_dtype}
        for name in COVARIANCE_FACTOR_NAMES:
            factor = load_factors(model, name, tracked_module_names, cpu=cpu, dtype=dtypes.get(name))
            if len(factor) == 0:
                raise ValueError(f"Factor `{name}` has not been computed.")
            saved[name] = factor
    model.zero_grad(set_to_none=True)
    set_attention_mask(model, None)
    set_gradient_scale(model, 1.0)
    set_mode(model, ModuleMode.DEFAULT, release_memory=True)
    state.wait_for_everyone()
    return num_data_processed, saved


def fit_covariance_matrices_with_loader(*args, **kwargs) -> Tuple[torch.Tensor, FACTOR_TYPE]:
    """Stage entry point (signature of ``_fit_covariance_matrices_with_loader_impl``); runs the loop
    with the cyclic GC paused (see ``utils.state.paused_gc``)."""
    with paused_gc():
        return _fit_covariance_matrices_with_loader_impl(*args, **kwargs)
