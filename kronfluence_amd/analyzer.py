"""``prepare_model`` and ``Analyzer``: the user-facing API (reference ``analyzer.py:20-242`` and the
orchestration subset of ``computer/{computer,factor_computer,score_computer}.py`` that the EK-FAC
hot path needs: data/module partitions with aggregation, automatic batch-size search, skip-if-exists, the same
``influence_results/<analysis>/factors_<name>/*.safetensors`` layout).

There is no CPU mode: every stage runs on an MI355X through ``libkronfluence_hip.so``.
"""

from __future__ import annotations

import logging
import os
import time
from pathlib import Path
import dataclasses
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import nn
from torch.utils import data
from torch.utils.data import DistributedSampler, SequentialSampler

from kronfluence_amd.arguments import FactorArguments, ScoreArguments
from kronfluence_amd.factor.config import FactorConfig
from kronfluence_amd.factor.covariance import (
    covariance_matrices_exist, fit_covariance_matrices_with_loader, load_covariance_matrices, save_covariance_matrices,
)
from kronfluence_amd.factor.eigen import (
    eigendecomposition_exist, fit_lambda_matrices_with_loader, lambda_matrices_exist, load_eigendecomposition,
    load_lambda_matrices, perform_eigendecomposition, save_eigendecomposition, save_lambda_matrices,
)
from kronfluence_amd.module.tracked_module import TrackedModule
from kronfluence_amd.module.utils import get_tracked_module_names, make_modules_partition, wrap_tracked_modules
from kronfluence_amd.score.pairwise import (
    compute_pairwise_query_aggregated_scores_with_loaders, compute_pairwise_scores_with_loaders, load_pairwise_scores, pairwise_scores_exist, save_pairwise_scores,
)
from kronfluence_amd.score.query_exchange import (
    backend_name, layer_shapes, mark_replicated, plan_query_exchange, probe_rows, requested_mode,
)
from kronfluence_amd.score.self import (
    compute_self_measurement_scores_with_loaders, compute_self_scores_with_loaders, load_self_scores, save_self_scores,
    self_scores_exist,
)
from kronfluence_amd.task import Task
from kronfluence_amd.utils.constants import FACTOR_SAVE_PREFIX, FACTOR_TYPE, SCORE_SAVE_PREFIX, SCORE_TYPE
from kronfluence_amd.utils.dataset import (
    DataLoaderKwargs, DistributedEvalSampler, DistributedSamplerWithStack, find_executable_batch_size, make_indices_partition,
    send_to_device,
)
from kronfluence_amd.utils.exceptions import FactorsNotFoundError, TrackedModuleNotFoundError
from kronfluence_amd.utils.save import load_file as load_safetensors
from kronfluence_amd.utils.save import load_json, save_json
from kronfluence_amd.utils.state import State, release_memory


def prepare_model(model: nn.Module, task: Task) -> nn.Module:
    """Freezes every parameter and buffer, switches to eval mode and installs ``TrackedModule``
    wrappers (reference ``analyzer.py:20-45``)."""
    model.eval()
    for tensor in list(model.parameters()) + list(model.buffers()):
        tensor.requires_grad = False
    return wrap_tracked_modules(model=model, task=task)


@dataclass
class _PartitionPlan:
    partitioned: bool
    ranges: List[Tuple[int, int]]
    modules: List[List[str]]
    data_targets: List[int]
    module_targets: List[int]

    def cells(self):
        for d in self.data_targets:
            for m in self.module_targets:
                yield (d, m) if self.partitioned else None, self.ranges[d], self.modules[m]


class Analyzer:
    def __init__(self, analysis_name: str, model: nn.Module, task: Task, cpu: bool = False,
                 log_level: Optional[int] = None, log_main_process_only: bool = True, profile: bool = False,
                 disable_tqdm: bool = False, output_dir: str = "./influence_results",
                 disable_model_save: bool = True) -> None:
        del log_main_process_only
        self._require_gpu(cpu)
        self.name, self.task, self.disable_tqdm, self.profile = analysis_name, task, disable_tqdm, profile
        self.state = State(cpu=False)
        self.logger = logging.getLogger(f"kronfluence_amd.{analysis_name}")
        if log_level is not None:
            self.logger.setLevel(log_level)
        if not any(isinstance(m, TrackedModule) for m in model.modules()):
            raise TrackedModuleNotFoundError(
                f"No `TrackedModule` found in model. Call `prepare_model` before initializing `Analyzer`."
            )
        self.model = model.to(self.state.device)
        self.output_dir = Path(output_dir).joinpath(analysis_name).resolve()
        if self.state.is_main_process:
            os.makedirs(self.output_dir, exist_ok=True)
        if not disable_model_save:
            # every rank compares (read-only) with an existing file, so a mismatch raises on ALL of them instead of leaving the others
            # blocked in the barrier below (the reference raises on rank 0 only); rank 0 alone writes a new file
            existed = (self.output_dir / "model.safetensors").exists()
            self.state.wait_for_everyone()
            if existed or self.state.is_main_process:
                self._save_model(write=self.state.is_main_process)
        self._dataloader_params = DataLoaderKwargs()
        self.timings: Dict[str, float] = {}
        self.state.wait_for_everyone()

    # -- helpers -----------------------------------------------------------------------------------
    def _save_model(self, write: bool = True) -> None:
        """``disable_model_save=False`` (reference analyzer.py:107-143): the first Analyzer of an output directory stores the model's
        state dict as ``model.safetensors``; every later one compares its model with that file and refuses to go on with a
        different one (factors and scores under this name belong to the stored model)."""
        from safetensors.torch import save_file

        from kronfluence_amd.utils.save import verify_models_equivalence

        path = self.output_dir / "model.safetensors"
        from torch.nn.parallel import DataParallel, DistributedDataParallel

        model = self.model.module if isinstance(self.model, (DataParallel, DistributedDataParallel)) else self.model
        state_dict = model.state_dict()
        if path.exists():
            if not verify_models_equivalence(load_safetensors(path), state_dict):
                message = (f"Detected a difference between the current model and the one saved at `{path}`. "
                           "Consider using a different `analysis_name` to avoid conflicts.")
                self.logger.error(message)
                raise ValueError(message)
            self.logger.info(f"Found existing saved model at `{path}`.")
            return
        if not write:
            return
        save_file({key: value.detach().to("cpu").clone().contiguous() for key, value in state_dict.items()}, str(path))
        self.logger.info(f"Saved model at `{path}`.")

    @staticmethod
    def _require_gpu(cpu: bool) -> None:
        if cpu:
            raise RuntimeError("`cpu=True` is not available: the MI355X-native engine has no CPU path.")
        if not torch.cuda.is_available():
            raise RuntimeError("No MI355X visible (torch.cuda.is_available() is False); there is no CPU fallback.")

    def set_dataloader_kwargs(self, dataloader_kwargs: DataLoaderKwargs) -> None:
        self._dataloader_params = dataloader_kwargs

    def factors_output_dir(self, factors_name: str) -> Path:
        return (self.output_dir / (FACTOR_SAVE_PREFIX + factors_name)).resolve()

    def scores_output_dir(self, scores_name: str) -> Path:
        return (self.output_dir / (SCORE_SAVE_PREFIX + scores_name)).resolve()

    def _device_sync(self) -> None:
        if self.state.device.type == "cuda":
            torch.cuda.synchronize(self.state.device)

    def _write_profile_summary(self, name: str) -> None:
        """``profile=True``: append the stage wall times measured so far (device-synchronised, see ``_timed``) to
        ``<output_dir>/profiler_output/<name>_summary_rank_<r>.txt`` -- the counterpart of the reference's profiler
        summaries (``computer/computer.py:323-333``), reduced to what the hot path needs: one line per stage."""
        if not self.profile or not self.timings:
            return
        directory = self.output_dir / "profiler_output"
        os.makedirs(directory, exist_ok=True)
        total = sum(self.timings.values())
        lines = [f"{'Action':<32}|{'Total time (s)':>16} |{'Percentage %':>14}"]
        for label, seconds in sorted(self.timings.items(), key=lambda item: -item[1]):
            lines.append(f"{label:<32}|{seconds:>16.4f} |{100.0 * seconds / total:>14.2f}")
        with open(directory / f"{name}_summary_rank_{self.state.process_index}.txt", "w", encoding="utf-8") as handle:
            handle.write("\n".join(lines) + "\n")

    def _timed(self, label: str):
        analyzer = self

        class _Timer:
            def __enter__(self):
                analyzer._device_sync()
                self.t0 = time.perf_counter()

            def __exit__(self, *exc):
                analyzer._device_sync()
                analyzer.timings[label] = analyzer.timings.get(label, 0.0) + time.perf_counter() - self.t0

        return _Timer()

    def _get_dataloader(self, dataset: data.Dataset, per_device_batch_size: int, dataloader_params: Dict,
                        indices: Optional[Sequence[int]] = None, allow_duplicates: bool = False,
                        stack: bool = False, replicate: bool = False) -> data.DataLoader:
        """Sampler choice as reference ``computer/computer.py:193-239``; ``replicate``: every rank iterates the whole dataset in
        order (the replicated query side of ``score/query_exchange.py``)."""
        if indices is not None:
            dataset = data.Subset(dataset=dataset, indices=indices)
        if self.state.use_distributed and replicate:
            sampler = SequentialSampler(dataset)
        elif self.state.use_distributed and not allow_duplicates:
            sampler = DistributedEvalSampler(dataset, num_replicas=self.state.num_processes, rank=self.state.process_index)
        elif self.state.use_distributed and stack:
            sampler = DistributedSamplerWithStack(dataset, num_replicas=self.state.num_processes, rank=self.state.process_index)
        elif self.state.use_distributed:
            sampler = DistributedSampler(dataset, num_replicas=self.state.num_processes, rank=self.state.process_index,
                                         shuffle=False, drop_last=False)
        else:
            sampler = SequentialSampler(dataset)
        return data.DataLoader(dataset=dataset, batch_size=per_device_batch_size, sampler=sampler, drop_last=False,
                               shuffle=False, **dataloader_params)

    def _reset_memory(self) -> None:
        from kronfluence_amd.module.tracked_module import ModuleMode
        from kronfluence_amd.module.utils import set_mode

        self.model.zero_grad(set_to_none=True)
        set_mode(self.model, ModuleMode.DEFAULT, release_memory=True)
        release_memory()

    def _find_executable_batch_size(self, probe, start: int) -> int:
        """Largest batch size (halving from ``start``) for which ``probe(batch_size)`` -- one batch through the stage --
        does not run out of HBM (reference ``utils/dataset.py:66-101``, ``factor_computer.py:110-157``)."""
        if self.state.use_distributed:
            raise NotImplementedError("Automatic batch size search is not supported for multi-GPU setting. "
                                      "Please manually configure the batch size by passing in `per_device_batch_size`.")
        def attempt(batch_size: int) -> None:
            self._reset_memory()
            probe(batch_size)

        batch_size = find_executable_batch_size(attempt, start)
        self._reset_memory()
        self.logger.info(f"Executable batch size determined: {batch_size}.")
        return batch_size

    def _partition_plan(self, total_examples: int, data_partitions: int, module_partitions: int,
                        target_data_partitions, target_module_partitions) -> "_PartitionPlan":
        """Index ranges / module-name lists of every partition and the ones this call computes
        (reference ``computer/computer.py:249-316``; same error rules)."""
        partitioned = not (data_partitions == 1 and module_partitions == 1)
        if not partitioned and (target_data_partitions is not None or target_module_partitions is not None):
            raise ValueError("`target_data_partitions` or `target_module_partitions` were specified, while the "
                             "arguments did not expect any data and module partition.")
        if total_examples < data_partitions:
            raise ValueError(f"Data partition size ({data_partitions}) exceeds total data points ({total_examples}). "
                             "Please reduce the data partition size.")
        names = get_tracked_module_names(self.model)
        if len(names) < module_partitions:
            raise ValueError(f"Module partition size ({module_partitions}) exceeds total tracked modules "
                             f"({len(names)}). Please reduce the module partition size.")
        if total_examples // data_partitions < self.state.num_processes:
            raise ValueError("The number of processes are larger than the data points per partition. "
                             "Try reducing the number of processes or the data partitions.")

        def targets(wanted, count: int, what: str) -> List[int]:
            if wanted is None:
                return list(range(count))
            wanted = [wanted] if isinstance(wanted, int) else list(wanted)
            for index in wanted:
                if index < 0 or index >= count:
                    raise ValueError(f"Invalid {what} partition {index}. Must be in range [0, {count}).")
            return wanted

        return _PartitionPlan(
            partitioned=partitioned,
            ranges=make_indices_partition(total_data_examples=total_examples, partition_size=data_partitions),
            modules=make_modules_partition(total_module_names=names, partition_size=module_partitions),
            data_targets=targets(target_data_partitions, data_partitions, "data"),
            module_targets=targets(target_module_partitions, module_partitions, "module"),
        )

    @staticmethod
    def _arguments_from_json(cls, path: Path):
        """Rebuilds an ``Arguments`` dataclass from its JSON form (dtypes are stored as ``"torch.float32"`` strings)."""
        return cls(**{k: (getattr(torch, v.split(".")[1]) if isinstance(v, str) and v.startswith("torch.") else v)
                      for k, v in load_json(path).items()})

    def _stored_factor_args(self, factors_name: str, aggregating: bool = False) -> FactorArguments:
        stored = self.load_factor_args(factors_name)
        if stored is None and aggregating:   # the reference's aggregate_* raise ValueError here (factor_computer.py:361-368)
            raise ValueError(f"Arguments for factors with name `{factors_name}` was not found when trying to aggregate factors.")
        if stored is None:
            raise FactorsNotFoundError(f"Factors with name `{factors_name}` not found at `{self.factors_output_dir(factors_name)}`.")
        return stored

    def _stored_score_args(self, scores_name: str) -> ScoreArguments:
        stored = self.load_score_args(scores_name)
        if stored is None:
            raise ValueError(f"Arguments for scores with name `{scores_name}` was not found when trying to aggregate influence scores.")
        return stored

    @torch.no_grad()
    def _aggregate_factors(self, factors_name: str, data_partitions: int, module_partitions: int, exist_fnc, load_fnc,
                           save_fnc, metadata: Optional[Dict[str, str]] = None) -> Optional[FACTOR_TYPE]:
        """Sum over data partitions, union over module partitions (reference ``factor_computer.py:57-108``);
        nothing happens until every partition file exists."""
        out = self.factors_output_dir(factors_name)
        if not out.exists():
            raise FileNotFoundError(f"Factors directory `{out}` not found when trying to aggregate factors.")
        grid = [(i, j) for i in range(data_partitions) for j in range(module_partitions)]
        if not all(exist_fnc(out, partition=cell) for cell in grid):
            return None
        total: FACTOR_TYPE = {}
        for cell in grid:
            for factor_name, per_module in load_fnc(out, partition=cell).items():
                held = total.setdefault(factor_name, {})
                for module_name, tensor in per_module.items():
                    if module_name in held:
                        held[module_name].add_(tensor)
                    else:
                        held[module_name] = tensor.clone()
        save_fnc(out, total, metadata=metadata)  # the per-partition files carry the same metadata
        return total

    @torch.no_grad()
    def _aggregate_scores(self, scores_name: str, exist_fnc, load_fnc, save_fnc, dim: int) -> Optional[SCORE_TYPE]:
        """Module partitions add, data partitions concatenate along the train axis -- or add, when the train
        gradients were aggregated (reference ``score_computer.py:77-139``)."""
        score_args = self._stored_score_args(scores_name)   # ValueError when nothing was computed under this name, as the reference
        out = self.scores_output_dir(scores_name)
        if not out.exists():
            raise FileNotFoundError(f"Scores directory `{out}` not found when trying to aggregate scores.")
        grid = [(i, j) for i in range(score_args.data_partitions) for j in range(score_args.module_partitions)]
        if not all(exist_fnc(out, partition=cell) for cell in grid):
            return None
        total: SCORE_TYPE = {}
        for i in range(score_args.data_partitions):
            block: SCORE_TYPE = {}
            for j in range(score_args.module_partitions):
                for key, tensor in load_fnc(out, partition=(i, j)).items():
                    if key in block:
                        block[key].add_(tensor)
                    else:
                        block[key] = tensor.clone()
            for key, tensor in block.items():
                if key not in total:
                    total[key] = tensor
                elif score_args.aggregate_train_gradients:
                    total[key].add_(tensor)
                else:
                    total[key] = torch.cat((total[key], tensor), dim=dim)
        save_fnc(out, total, metadata=score_args.to_str_dict())
        return total

    def _save_arguments(self, path: Path, arguments, overwrite: bool) -> None:
        """Every rank compares (read-only) so that a mismatch raises everywhere -- a rank-0-only ``ValueError`` would
        leave the other ranks waiting in the stage's first collective; only the main process writes."""
        exists = path.exists()
        self.state.wait_for_everyone()  # nobody inspects the directory after rank 0 has started writing into it
        if exists and not overwrite:
            stored = load_json(path)
            if stored != arguments.to_dict():
                raise ValueError(f"Arguments stored at `{path}` differ from the ones provided; pass "
                                 f"`overwrite_output_dir=True` or use a different name.")
        elif self.state.is_main_process:
            save_json(arguments.to_dict(), path)

    def _save_dataset_metadata(self, name: str, dataset: data.Dataset, output_dir: Path,
                               indices: Optional[Sequence[int]] = None, overwrite: bool = False) -> None:
        """``<name>_dataset_metadata.json`` next to the results (type, size, index subset): a later call on the same
        output directory with a different dataset is an error unless overwriting (reference ``computer.py:160-191``).
        Compared on every rank, written by the main process (see ``_save_arguments``)."""
        path = output_dir / f"{name}_dataset_metadata.json"
        metadata = {"type": type(dataset).__name__, "dataset_size": len(dataset),
                    "indices": None if indices is None else list(indices)}
        exists = path.exists()
        self.state.wait_for_everyone()
        if exists and not overwrite:
            stored = load_json(path)
            if stored != metadata:
                raise ValueError("Attempting to use the dataset that differs from the one already saved. Please set "
                                 f"`overwrite_output_dir=True` to overwrite existing experiment.\nNew metadata: {metadata}."
                                 f"\nSaved metadata: {stored}.")
        elif self.state.is_main_process:
            save_json(metadata, path)

    # -- factors -----------------------------------------------------------------------------------
    def fit_covariance_matrices(self, factors_name: str, dataset: data.Dataset,
                                per_device_batch_size: Optional[int] = None,
                                initial_per_device_batch_size_attempt: int = 4096,
                                dataloader_kwargs: Optional[DataLoaderKwargs] = None,
                                factor_args: Optional[FactorArguments] = None,
                                target_data_partitions: Optional[Sequence[int]] = None,
                                target_module_partitions: Optional[Sequence[int]] = None,
                                overwrite_output_dir: bool = False) -> None:
        factor_args = factor_args or FactorArguments()
        out = self.factors_output_dir(factors_name)
        if self.state.is_main_process:
            os.makedirs(out, exist_ok=True)
        self.state.wait_for_everyone()
        if covariance_matrices_exist(out) and not overwrite_output_dir:
            return
        self._save_arguments(out / "factor_arguments.json", factor_args, overwrite_output_dir)
        if not FactorConfig.CONFIGS[factor_args.strategy].requires_covariance_matrices:
            return
        self._save_dataset_metadata("covariance", dataset, out, overwrite=overwrite_output_dir)
        batch_size = per_device_batch_size
        total = len(dataset) if factor_args.covariance_max_examples is None else min(factor_args.covariance_max_examples, len(dataset))
        plan = self._partition_plan(total, factor_args.covariance_data_partitions, factor_args.covariance_module_partitions,
                                    target_data_partitions, target_module_partitions)
        params = (dataloader_kwargs or self._dataloader_params).to_dict()
        for partition, (start, end), module_names in plan.cells():
            if covariance_matrices_exist(out, partition) and not overwrite_output_dir:
                continue
            if batch_size is None:
                def probe(size: int, names=module_names) -> None:
                    loader = self._get_dataloader(dataset, size, params, indices=list(range(size)), allow_duplicates=True)
                    fit_covariance_matrices_with_loader(self.model, self.state, self.task, loader, factor_args,
                                                        tracked_module_names=names)

                batch_size = self._find_executable_batch_size(
                    probe, min(initial_per_device_batch_size_attempt, total // factor_args.covariance_data_partitions))
            loader = self._get_dataloader(dataset, batch_size, params, indices=list(range(start, end)), allow_duplicates=False)
            with self._timed("fit_covariance"):
                _, factors = fit_covariance_matrices_with_loader(self.model, self.state, self.task, loader, factor_args,
                                                                 tracked_module_names=module_names)
            if self.state.is_main_process:
                save_covariance_matrices(out, factors, partition=partition, metadata=factor_args.to_str_dict())
            self.state.wait_for_everyone()
        if plan.partitioned:
            if self.state.is_main_process:
                self.aggregate_covariance_matrices(factors_name)
            self.state.wait_for_everyone()

    def aggregate_covariance_matrices(self, factors_name: str) -> None:
        """Aggregates the partitioned covariance files once all of them exist (reference ``factor_computer.py:350-378``)."""
        factor_args = self._stored_factor_args(factors_name, aggregating=True)
        self._aggregate_factors(factors_name, factor_args.covariance_data_partitions, factor_args.covariance_module_partitions,
                                covariance_matrices_exist, load_covariance_matrices, save_covariance_matrices,
                                metadata=factor_args.to_str_dict())

    def perform_eigendecomposition(self, factors_name: str, factor_args: Optional[FactorArguments] = None,
                                   overwrite_output_dir: bool = False,
                                   load_from_factors_name: Optional[str] = None) -> None:
        factor_args = factor_args or FactorArguments()
        out = self.factors_output_dir(factors_name)
        if self.state.is_main_process:
            os.makedirs(out, exist_ok=True)
        self.state.wait_for_everyone()
        if eigendecomposition_exist(out) and not overwrite_output_dir:
            return
        self._save_arguments(out / "factor_arguments.json", factor_args, overwrite_output_dir)
        if not FactorConfig.CONFIGS[factor_args.strategy].requires_eigendecomposition:
            return
        source = self.factors_output_dir(load_from_factors_name) if load_from_factors_name else out
        if not covariance_matrices_exist(source):
            raise FactorsNotFoundError(f"Covariance matrices not found at `{source}`. "
                                       f"To perform eigendecomposition, call `fit_covariance_matrices` first.")
        covariance = load_covariance_matrices(source)
        if load_from_factors_name is not None:
            # reference factor_computer.py:434-444: the borrowed covariances become part of THIS name (load_covariance_matrices of
            # it returns them), together with the arguments they were fitted with
            if self.state.is_main_process:
                save_covariance_matrices(out, covariance)
            self._save_arguments(out / "factor_loaded_covariance_arguments.json", self._stored_factor_args(load_from_factors_name), True)
            self.state.wait_for_everyone()
        with self._timed("perform_eigendecomposition"):
            eigen = perform_eigendecomposition(covariance, self.model, self.state, factor_args)
        if self.state.is_main_process:
            save_eigendecomposition(out, eigen, metadata=factor_args.to_str_dict())
        self.state.wait_for_everyone()

    def fit_lambda_matrices(self, factors_name: str, dataset: data.Dataset, per_device_batch_size: Optional[int] = None,
                            initial_per_device_batch_size_attempt: int = 4096,
                            dataloader_kwargs: Optional[DataLoaderKwargs] = None,
                            factor_args: Optional[FactorArguments] = None,
                            target_data_partitions: Optional[Sequence[int]] = None,
                            target_module_partitions: Optional[Sequence[int]] = None,
                            overwrite_output_dir: bool = False,
                            load_from_factors_name: Optional[str] = None) -> None:
        factor_args = factor_args or FactorArguments()
        out = self.factors_output_dir(factors_name)
        if self.state.is_main_process:
            os.makedirs(out, exist_ok=True)
        self.state.wait_for_everyone()
        if lambda_matrices_exist(out) and not overwrite_output_dir:
            return
        self._save_arguments(out / "factor_arguments.json", factor_args, overwrite_output_dir)
        config = FactorConfig.CONFIGS[factor_args.strategy]
        if not config.requires_lambda_matrices:
            return
        self._save_dataset_metadata("lambda", dataset, out, overwrite=overwrite_output_dir)
        eigen = None
        if config.requires_eigendecomposition_for_lambda:
            source = self.factors_output_dir(load_from_factors_name) if load_from_factors_name else out
            if not eigendecomposition_exist(source):
                raise FactorsNotFoundError(f"Eigendecomposition results not found at `{source}`. "
                                           f"To fit Lambda matrices, call `perform_eigendecomposition` first.")
            eigen = load_eigendecomposition(source)
            if load_from_factors_name is not None:   # reference factor_computer.py:563-573
                if self.state.is_main_process:
                    save_eigendecomposition(out, eigen)
                self._save_arguments(out / "factor_loaded_eigendecomposition_arguments.json",
                                     self._stored_factor_args(load_from_factors_name), True)
                self.state.wait_for_everyone()
        batch_size = per_device_batch_size
        total = len(dataset) if factor_args.lambda_max_examples is None else min(factor_args.lambda_max_examples, len(dataset))
        plan = self._partition_plan(total, factor_args.lambda_data_partitions, factor_args.lambda_module_partitions,
                                    target_data_partitions, target_module_partitions)
        params = (dataloader_kwargs or self._dataloader_params).to_dict()
        for partition, (start, end), module_names in plan.cells():
            if lambda_matrices_exist(out, partition) and not overwrite_output_dir:
                continue
            if batch_size is None:
                def probe(size: int, names=module_names) -> None:
                    loader = self._get_dataloader(dataset, size, params, indices=list(range(size)), allow_duplicates=True)
                    fit_lambda_matrices_with_loader(self.model, self.state, self.task, loader, factor_args, eigen,
                                                    tracked_module_names=names)

                batch_size = self._find_executable_batch_size(
                    probe, min(initial_per_device_batch_size_attempt, total // factor_args.lambda_data_partitions))
            loader = self._get_dataloader(dataset, batch_size, params, indices=list(range(start, end)), allow_duplicates=False)
            with self._timed("fit_lambda"):
                _, factors = fit_lambda_matrices_with_loader(self.model, self.state, self.task, loader, factor_args, eigen,
                                                             tracked_module_names=module_names)
            if self.state.is_main_process:
                save_lambda_matrices(out, factors, partition=partition, metadata=factor_args.to_str_dict())
            self.state.wait_for_everyone()
        if plan.partitioned:
            if self.state.is_main_process:
                self.aggregate_lambda_matrices(factors_name)
            self.state.wait_for_everyone()

    def aggregate_lambda_matrices(self, factors_name: str) -> None:
        """Aggregates the partitioned Lambda files once all of them exist (reference ``factor_computer.py:704-732``)."""
        factor_args = self._stored_factor_args(factors_name, aggregating=True)
        self._aggregate_factors(factors_name, factor_args.lambda_data_partitions, factor_args.lambda_module_partitions,
                                lambda_matrices_exist, load_lambda_matrices, save_lambda_matrices,
                                metadata=factor_args.to_str_dict())

    def fit_all_factors(self, factors_name: str, dataset: data.Dataset, per_device_batch_size: Optional[int] = None,
                        initial_per_device_batch_size_attempt: int = 4096,
                        dataloader_kwargs: Optional[DataLoaderKwargs] = None,
                        factor_args: Optional[FactorArguments] = None, overwrite_output_dir: bool = False) -> None:
        """Covariance -> eigendecomposition -> Lambda (reference ``analyzer.py:144-195``)."""
        common = dict(factors_name=factors_name, factor_args=factor_args, overwrite_output_dir=overwrite_output_dir)
        self.fit_covariance_matrices(dataset=dataset, per_device_batch_size=per_device_batch_size,
                                     initial_per_device_batch_size_attempt=initial_per_device_batch_size_attempt,
                                     dataloader_kwargs=dataloader_kwargs, **common)
        self.perform_eigendecomposition(**common)
        self.fit_lambda_matrices(dataset=dataset, per_device_batch_size=per_device_batch_size,
                                 initial_per_device_batch_size_attempt=initial_per_device_batch_size_attempt,
                                 dataloader_kwargs=dataloader_kwargs, **common)
        self._write_profile_summary(f"factors_{factors_name}")

    def load_factor_args(self, factors_name: str) -> Optional[FactorArguments]:
        """The ``FactorArguments`` the factors were fitted with (reference ``computer/computer.py:335-341``)."""
        path = self.factors_output_dir(factors_name) / "factor_arguments.json"
        return self._arguments_from_json(FactorArguments, path) if path.exists() else None

    def load_score_args(self, scores_name: str) -> Optional[ScoreArguments]:
        """The ``ScoreArguments`` the scores were computed with (reference ``computer/computer.py:371-377``)."""
        path = self.scores_output_dir(scores_name) / "score_arguments.json"
        return self._arguments_from_json(ScoreArguments, path) if path.exists() else None

    @staticmethod
    def load_file(path: Union[str, Path]) -> Dict[str, torch.Tensor]:
        """Loads one ``.safetensors`` file of factors or scores (reference ``analyzer.py:199-220``)."""
        path = Path(path).resolve() if isinstance(path, str) else path
        if not path.exists():
            raise FileNotFoundError(f"File not found: {path}.")
        return load_safetensors(path)

    @staticmethod
    def get_module_summary(model: nn.Module) -> str:
        """Names and reprs of the leaf modules that own parameters -- the candidates for
        ``Task.get_influence_tracked_modules`` (reference ``analyzer.py:222-242``)."""
        lines = ["==Model Summary=="]
        for name, module in model.named_modules():
            if any(True for _ in module.children()) or not any(True for _ in module.parameters()):
                continue
            lines.append(f"Module Name: `{name}`, Module: {repr(module)}")
        return "\n".join(lines)

    def load_covariance_matrices(self, factors_name: str) -> Optional[FACTOR_TYPE]:
        out = self.factors_output_dir(factors_name)
        return load_covariance_matrices(out) if covariance_matrices_exist(out) else None

    def load_eigendecomposition(self, factors_name: str) -> Optional[FACTOR_TYPE]:
        out = self.factors_output_dir(factors_name)
        return load_eigendecomposition(out) if eigendecomposition_exist(out) else None

    def load_lambda_matrices(self, factors_name: str) -> Optional[FACTOR_TYPE]:
        out = self.factors_output_dir(factors_name)
        return load_lambda_matrices(out) if lambda_matrices_exist(out) else None

    def load_all_factors(self, factors_name: str) -> FACTOR_TYPE:
        """Everything the strategy needs for preconditioning (reference ``computer/computer.py:387-434``)."""
        stored = self.load_factor_args(factors_name)
        if stored is None:
            raise FileNotFoundError(f"Factors with name `{factors_name}` was not found at `{self.factors_output_dir(factors_name)}`.")
        config = FactorConfig.CONFIGS[stored.strategy]
        loaded: FACTOR_TYPE = {}
        for required, load, what in (
                (config.requires_covariance_matrices_for_precondition, self.load_covariance_matrices, "covariance matrices"),
                (config.requires_eigendecomposition_for_precondition, self.load_eigendecomposition, "Eigendecomposition results"),
                (config.requires_lambda_matrices_for_precondition, self.load_lambda_matrices, "Lambda matrices")):
            if not required:
                continue
            factors = load(factors_name)
            if factors is None:
                raise FactorsNotFoundError(f"Strategy `{stored.strategy}` requires {what}. However, the {what} were not found.")
            loaded.update(factors)
        return loaded

    # -- scores ------------------------------------------------------------------------------------
    def compute_pairwise_scores(self, scores_name: str, factors_name: str, query_dataset: data.Dataset,
                                train_dataset: data.Dataset, per_device_query_batch_size: int,
                                per_device_train_batch_size: Optional[int] = None,
                                initial_per_device_train_batch_size_attempt: int = 4096,
                                query_indices: Optional[Sequence[int]] = None,
                                train_indices: Optional[Sequence[int]] = None,
                                dataloader_kwargs: Optional[DataLoaderKwargs] = None,
                                score_args: Optional[ScoreArguments] = None,
                                target_data_partitions: Optional[Sequence[int]] = None,
                                target_module_partitions: Optional[Sequence[int]] = None,
                                overwrite_output_dir: bool = False) -> Optional[SCORE_TYPE]:
        score_args = score_args or ScoreArguments()
        out = self.scores_output_dir(scores_name)
        if self.state.is_main_process:
            os.makedirs(out, exist_ok=True)
        self.state.wait_for_everyone()
        if pairwise_scores_exist(out) and not overwrite_output_dir:
            return self.load_pairwise_scores(scores_name)
        factor_args = self._stored_factor_args(factors_name)
        if score_args.compute_per_token_scores and (score_args.aggregate_train_gradients or factor_args.has_shared_parameters
                                                    or self.task.enable_post_process_per_sample_gradient):
            # reference score_computer.py:287-308: token-wise scores are silently disabled in these configurations
            self.logger.warning("Token-wise influence computation is not compatible with this configuration; "
                                "disabling `compute_per_token_scores`.")
            score_args = dataclasses.replace(score_args, compute_per_token_scores=False)
        self._save_arguments(out / "score_arguments.json", score_args, overwrite_output_dir)
        self._save_arguments(out / "factor_arguments.json", factor_args, overwrite_output_dir)   # which factors these scores used
        loaded = self.load_all_factors(factors_name)
        if not loaded and FactorConfig.CONFIGS[factor_args.strategy].requires_lambda_matrices_for_precondition:
            raise FactorsNotFoundError(f"Factors with name `{factors_name}` are incomplete.")
        params = (dataloader_kwargs or self._dataloader_params).to_dict()
        train_batch = per_device_train_batch_size
        self._save_dataset_metadata("query", query_dataset, out, query_indices, overwrite_output_dir)
        self._save_dataset_metadata("train", train_dataset, out, train_indices, overwrite_output_dir)
        if query_indices is not None:
            query_dataset = data.Subset(dataset=query_dataset, indices=query_indices)
        if train_indices is not None:
            train_dataset = data.Subset(dataset=train_dataset, indices=train_indices)
        plan = self._partition_plan(len(train_dataset), score_args.data_partitions, score_args.module_partitions,
                                    target_data_partitions, target_module_partitions)
        scores = None
        stage = (compute_pairwise_query_aggregated_scores_with_loaders if score_args.aggregate_query_gradients
                 else compute_pairwise_scores_with_loaders)
        for partition, (start, end), module_names in plan.cells():
            if pairwise_scores_exist(out, partition) and not overwrite_output_dir:
                continue
            if train_batch is None:
                def probe(size: int, names=module_names) -> None:
                    # one query batch against one train batch of the candidate size (score_computer.py:141-215)
                    q_size = min(per_device_query_batch_size, len(query_dataset))
                    ql = self._get_dataloader(query_dataset, q_size, params, indices=list(range(q_size)), allow_duplicates=True)
                    tl = self._get_dataloader(train_dataset, size, params, indices=list(range(size)), allow_duplicates=True,
                                              stack=True)
                    stage(loaded_factors=loaded, model=self.model, state=self.state, task=self.task, query_loader=ql,
                          per_device_query_batch_size=q_size, train_loader=tl, score_args=score_args,
                          factor_args=factor_args, tracked_module_names=names)

                train_batch = self._find_executable_batch_size(
                    probe, min(initial_per_device_train_batch_size_attempt, len(train_dataset) // score_args.data_partitions))
            replicate = self._replicate_queries(query_dataset, per_device_query_batch_size, params, score_args, factor_args, module_names)
            query_loader = mark_replicated(self._get_dataloader(query_dataset, per_device_query_batch_size, params,
                                                                allow_duplicates=not score_args.aggregate_query_gradients,
                                                                replicate=replicate), replicate)
            train_loader = self._get_dataloader(train_dataset, train_batch, params, indices=list(range(start, end)),
                                                allow_duplicates=not score_args.aggregate_train_gradients,
                                                stack=not score_args.aggregate_train_gradients)
            with self._timed("compute_pairwise_scores"):
                scores = stage(
                    loaded_factors=loaded, model=self.model, state=self.state, task=self.task, query_loader=query_loader,
                    per_device_query_batch_size=per_device_query_batch_size, train_loader=train_loader,
                    score_args=score_args, factor_args=factor_args, tracked_module_names=module_names)
            if self.state.is_main_process:
                save_pairwise_scores(out, scores, partition=partition, metadata=score_args.to_str_dict())
            self.state.wait_for_everyone()
        if plan.partitioned:
            scores = None
            if self.state.is_main_process:
                scores = self.aggregate_pairwise_scores(scores_name)
            self.state.wait_for_everyone()
        self._write_profile_summary(f"scores_{scores_name}_pairwise")
        return scores if self.state.is_main_process else None

    def _replicate_queries(self, query_dataset: data.Dataset, per_device_query_batch_size: int, params: Dict,
                           score_args: ScoreArguments, factor_args: FactorArguments, module_names: Optional[List[str]]) -> bool:
        """Multi-rank runs: whether every rank preconditions ALL queries itself instead of all-gathering them
        (``score/query_exchange.py``: ``KF_QUERY_EXCHANGE``, or in "auto" the bytes-against-flops plan from the layer shapes and the
        rows per sample one no-grad forward of a single query shows).  The decision is the same on every rank (shapes, sizes and
        the backend are) and is logged."""
        if not self.state.use_distributed or score_args.aggregate_query_gradients:
            return False
        mode = requested_mode()
        shapes = layer_shapes(self.model, module_names)
        rows = [1] * len(shapes)
        if mode == "auto":
            first = next(iter(self._get_dataloader(query_dataset, 1, params, indices=[0], replicate=True)))
            first = send_to_device(first, self.state.device)
            enable_amp = score_args.amp_dtype is not None

            def measure() -> None:
                with torch.autocast(device_type=self.state.device.type, enabled=enable_amp, dtype=score_args.amp_dtype):
                    self.task.compute_measurement(batch=first, model=self.model)
            rows = probe_rows(self.model, measure, module_names)
        plan = plan_query_exchange(shapes, rows, len(query_dataset), self.state.num_processes, score_dtype=score_args.score_dtype,
                                   precondition_dtype=score_args.precondition_dtype, low_rank=score_args.query_gradient_low_rank,
                                   backend=backend_name(), mode=mode)
        self.logger.info(f"Query exchange: {plan.mode} ({plan.reason}).")
        self.last_query_exchange_plan = plan
        return plan.mode == "replicate"

    def aggregate_pairwise_scores(self, scores_name: str) -> Optional[SCORE_TYPE]:
        """Aggregates the partitioned score files once all of them exist (reference ``score_computer.py:466-482``)."""
        return self._aggregate_scores(scores_name, pairwise_scores_exist, load_pairwise_scores, save_pairwise_scores, dim=1)

    def load_pairwise_scores(self, scores_name: str) -> Optional[SCORE_TYPE]:
        out = self.scores_output_dir(scores_name)
        return load_pairwise_scores(out) if pairwise_scores_exist(out) else None

    def compute_self_scores(self, scores_name: str, factors_name: str, train_dataset: data.Dataset,
                            per_device_train_batch_size: Optional[int] = None,
                            initial_per_device_train_batch_size_attempt: int = 4096,
                            train_indices: Optional[Sequence[int]] = None,
                            dataloader_kwargs: Optional[DataLoaderKwargs] = None,
                            score_args: Optional[ScoreArguments] = None,
                            target_data_partitions: Optional[Sequence[int]] = None,
                            target_module_partitions: Optional[Sequence[int]] = None,
                            overwrite_output_dir: bool = False) -> Optional[SCORE_TYPE]:
        """Self-influence scores ``[N]`` (reference ``computer/score_computer.py:558-773``)."""
        score_args = score_args or ScoreArguments()
        out = self.scores_output_dir(scores_name)
        if self.state.is_main_process:
            os.makedirs(out, exist_ok=True)
        self.state.wait_for_everyone()
        if self_scores_exist(out) and not overwrite_output_dir:
            return self.load_self_scores(scores_name)
        factor_args = self._stored_factor_args(factors_name)
        if (score_args.query_gradient_low_rank is not None or score_args.aggregate_query_gradients
                or score_args.aggregate_train_gradients or score_args.compute_per_token_scores):
            # reference score_computer.py:620-640: these options do not apply to self-influence and are switched off
            self.logger.warning("Low-rank queries, gradient aggregation and token-wise scores do not apply to "
                                "self-influence; disabling them.")
            score_args = dataclasses.replace(score_args, query_gradient_low_rank=None, aggregate_query_gradients=False,
                                             aggregate_train_gradients=False, compute_per_token_scores=False)
        self._save_arguments(out / "score_arguments.json", score_args, overwrite_output_dir)
        self._save_arguments(out / "factor_arguments.json", factor_args, overwrite_output_dir)
        loaded = self.load_all_factors(factors_name)
        params = (dataloader_kwargs or self._dataloader_params).to_dict()
        train_batch = per_device_train_batch_size
        self._save_dataset_metadata("train", train_dataset, out, train_indices, overwrite_output_dir)
        if train_indices is not None:
            train_dataset = data.Subset(dataset=train_dataset, indices=train_indices)
        plan = self._partition_plan(len(train_dataset), score_args.data_partitions, score_args.module_partitions,
                                    target_data_partitions, target_module_partitions)
        stage = (compute_self_measurement_scores_with_loaders if score_args.use_measurement_for_self_influence
                 else compute_self_scores_with_loaders)
        scores = None
        for partition, (start, end), module_names in plan.cells():
            if self_scores_exist(out, partition) and not overwrite_output_dir:
                continue
            if train_batch is None:
                def probe(size: int, names=module_names) -> None:
                    tl = self._get_dataloader(train_dataset, size, params, indices=list(range(size)), allow_duplicates=True,
                                              stack=True)
                    stage(loaded_factors=loaded, model=self.model, state=self.state, task=self.task, train_loader=tl,
                          score_args=score_args, factor_args=factor_args, tracked_module_names=names)

                train_batch = self._find_executable_batch_size(
                    probe, min(initial_per_device_train_batch_size_attempt, len(train_dataset) // score_args.data_partitions))
            train_loader = self._get_dataloader(train_dataset, train_batch, params, indices=list(range(start, end)),
                                                allow_duplicates=True, stack=True)
            with self._timed("compute_self_scores"):
                scores = stage(loaded_factors=loaded, model=self.model, state=self.state, task=self.task,
                               train_loader=train_loader, score_args=score_args, factor_args=factor_args,
                               tracked_module_names=module_names)
            if self.state.is_main_process:
                save_self_scores(out, scores, partition=partition, metadata=score_args.to_str_dict())
            self.state.wait_for_everyone()
        if plan.partitioned:
            scores = None
            if self.state.is_main_process:
                scores = self.aggregate_self_scores(scores_name)
            self.state.wait_for_everyone()
        self._write_profile_summary(f"scores_{scores_name}_self")
        return scores if self.state.is_main_process else None

    def aggregate_self_scores(self, scores_name: str) -> Optional[SCORE_TYPE]:
        """Aggregates the partitioned self-score files once all of them exist (reference ``score_computer.py:775-790``)."""
        return self._aggregate_scores(scores_name, self_scores_exist, load_self_scores, save_self_scores, dim=0)

    def load_self_scores(self, scores_name: str) -> Optional[SCORE_TYPE]:
        out = self.scores_output_dir(scores_name)
        return load_self_scores(out) if self_scores_exist(out) else None
