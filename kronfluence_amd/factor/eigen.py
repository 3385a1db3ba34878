"""Stage 2: eigendecomposition (reference ``factor/eigen.py:140-224``) and Lambda fitting
(``:345-462``), plus their safetensors layout (``:46-91, 227-290``)."""

from __future__ import annotations

import contextlib
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from safetensors.torch import load_file, save_file
from torch import autocast, nn
from torch.utils import data

from kronfluence_amd import ops
from kronfluence_amd.arguments import FactorArguments
from kronfluence_amd.factor.covariance import _loss_scale
from kronfluence_amd.module.tracked_module import ModuleMode
from kronfluence_amd.module.utils import (
    READ_ONLY_FACTORS,
    finalize_iteration, get_tracked_module_names, load_factors, set_factors, set_gradient_scale, set_mode, set_side_stream,
    synchronize_factors, update_factor_args,
)
from kronfluence_amd.task import Task
from kronfluence_amd.utils.comm import exchange
from kronfluence_amd.utils.constants import (
    ACTIVATION_COVARIANCE_MATRIX_NAME, ACTIVATION_EIGENVALUES_NAME, ACTIVATION_EIGENVECTORS_NAME,
    EIGENDECOMPOSITION_FACTOR_NAMES, FACTOR_TYPE, GRADIENT_COVARIANCE_MATRIX_NAME, GRADIENT_EIGENVALUES_NAME,
    GRADIENT_EIGENVECTORS_NAME, LAMBDA_FACTOR_NAMES, LAMBDA_MATRIX_NAME, NUM_ACTIVATION_COVARIANCE_PROCESSED,
    NUM_GRADIENT_COVARIANCE_PROCESSED,
)
from kronfluence_amd.utils.dataset import find_batch_size, send_to_device
from kronfluence_amd.utils.state import State, no_sync, paused_gc


def _eigh_streams() -> int:
    """``KF_EIGH_STREAMS`` (default 8; 1 under rocprofv3, whose tracer does not survive eight launching threads)."""
    value = os.environ.get("KF_EIGH_STREAMS", "8")
    try:
        return max(1, int(value))
    except ValueError:
        raise ValueError(f"KF_EIGH_STREAMS must be a positive integer, got {value!r}") from None


# concurrent eigenproblems (HIP streams / host threads) per rank: the in-LDS solve of a round is latency bound on a few dozen CUs, the
# other problems' streaming kernels fill the rest of the chip meanwhile
EIGH_STREAMS = _eigh_streams()


def eigendecomposition_save_path(output_dir: Path, factor_name: str) -> Path:
    assert factor_name in EIGENDECOMPOSITION_FACTOR_NAMES
    return output_dir / f"{factor_name}.safetensors"


def save_eigendecomposition(output_dir: Path, factors: FACTOR_TYPE, metadata: Optional[Dict[str, str]] = None) -> None:
    assert set(factors.keys()) == set(EIGENDECOMPOSITION_FACTOR_NAMES)
    for name in factors:
        save_file(tensors={k: v.contiguous() for k, v in factors[name].items()},
                  filename=str(eigendecomposition_save_path(output_dir, name)), metadata=metadata)


def load_eigendecomposition(output_dir: Path) -> FACTOR_TYPE:
    return {name: load_file(filename=str(eigendecomposition_save_path(output_dir, name)))
            for name in EIGENDECOMPOSITION_FACTOR_NAMES}


def eigendecomposition_exist(output_dir: Path) -> bool:
    return all(eigendecomposition_save_path(output_dir, name).exists() for name in EIGENDECOMPOSITION_FACTOR_NAMES)


def lambda_matrices_save_path(output_dir: Path, factor_name: str, partition=None) -> Path:
    assert factor_name in LAMBDA_FACTOR_NAMES
    if partition is not None:
        return output_dir / f"{factor_name}_data_partition{partition[0]}_module_partition{partition[1]}.safetensors"
    return output_dir / f"{factor_name}.safetensors"


def save_lambda_matrices(output_dir: Path, factors: FACTOR_TYPE, partition=None, metadata: Optional[Dict[str, str]] = None) -> None:
    assert set(factors.keys()) == set(LAMBDA_FACTOR_NAMES)
    for name in factors:
        save_file(tensors={k: v.contiguous() for k, v in factors[name].items()},
                  filename=str(lambda_matrices_save_path(output_dir, name, partition)), metadata=metadata)


def load_lambda_matrices(output_dir: Path, partition=None) -> FACTOR_TYPE:
    return {name: load_file(filename=str(lambda_matrices_save_path(output_dir, name, partition)))
            for name in LAMBDA_FACTOR_NAMES}


def lambda_matrices_exist(output_dir: Path, partition=None) -> bool:
    return all(lambda_matrices_save_path(output_dir, name, partition).exists() for name in LAMBDA_FACTOR_NAMES)


# perform_eigendecomposition solves covariance matrices that are the same up to summation order once (False: every matrix on its own)
DEDUPLICATE_COVARIANCES = True


@torch.no_grad()
def perform_eigendecomposition(covariance_factors: FACTOR_TYPE, model: nn.Module, state: State,
                               factor_args: FactorArguments, disable_tqdm: bool = False, cpu: bool = True,
                               release_covariances: bool = False) -> FACTOR_TYPE:
    """``eigh(0.5 (C + C^T) / count)`` in fp64 for both factors of every tracked layer, on the MI355X
    (``kf_eigh_f64``).  The 2L matrices are independent: with several ranks they are dealt
    round-robin and the results are exchanged by broadcast, instead of rank 0 doing all of them while
    the others wait at a barrier (reference ``factor_computer.py:449-470``).  Results are cast back
    to the covariance dtype and returned on the CPU (``eigen.py:214-219``), or left in HBM with ``cpu=False``.
    ``release_covariances``: every covariance matrix is dropped from ``covariance_factors`` as soon as its problem is solved --
    the streaming the wide configs need (Llama-3-8B: 85 GB of covariances next to 85 GB of eigenvectors)."""
    del disable_tqdm
    out: FACTOR_TYPE = {name: {} for name in EIGENDECOMPOSITION_FACTOR_NAMES}
    jobs = []
    for module_name in get_tracked_module_names(model):
        for cov_name, count_name, vec_name, val_name in (
            (ACTIVATION_COVARIANCE_MATRIX_NAME, NUM_ACTIVATION_COVARIANCE_PROCESSED, ACTIVATION_EIGENVECTORS_NAME,
             ACTIVATION_EIGENVALUES_NAME),
            (GRADIENT_COVARIANCE_MATRIX_NAME, NUM_GRADIENT_COVARIANCE_PROCESSED, GRADIENT_EIGENVECTORS_NAME,
             GRADIENT_EIGENVALUES_NAME),
        ):
            jobs.append((module_name, cov_name, count_name, vec_name, val_name))
    world = state.num_processes if (state.use_distributed and dist.is_initialized()) else 1
    # Layers that consume the SAME tensor have the same activation covariance (query / key / value projections of an attention block,
    # the gate / up projections of a SwiGLU MLP): the reference solves each -- from bit-identical matrices, so it gets identical
    # results.  Here such a matrix is solved once and its eigendecomposition shared: two accumulations of one input differ only in
    # the order of their fp32 atomics (relative 1e-7), so "the same" means equal counts and ||A - B||_F <= 1e-5 ||A||_F, screened by the
    # diagonals first.  Every rank sees the same all-reduced covariances, hence takes the same decisions.
    alias: Dict[int, int] = {}
    if DEDUPLICATE_COVARIANCES:
        seen: Dict[tuple, list] = {}
        for index, (module_name, cov_name, count_name, _vec, _val) in enumerate(jobs):
            cov = covariance_factors[cov_name][module_name]
            count = int(covariance_factors[count_name][module_name].item())
            key = (cov_name, tuple(cov.shape), cov.dtype, count)
            diag = cov.diagonal().to(device=state.device, dtype=torch.float64)
            for other in seen.setdefault(key, []):
                rep_diag, rep_index = other
                if float((diag - rep_diag).norm()) > 1e-5 * float(rep_diag.norm()):
                    continue
                rep = covariance_factors[cov_name][jobs[rep_index][0]]
                a64, b64 = cov.to(device=state.device, dtype=torch.float32), rep.to(device=state.device, dtype=torch.float32)
                if float((a64 - b64).norm()) <= 1e-5 * float(b64.norm()):
                    alias[index] = rep_index
                    break
            else:
                seen[key].append((diag, index))
    if world > 1 or (state.use_distributed and dist.is_initialized()):
        # the decisions above are floating-point threshold tests on all-reduced covariances: equal on every rank in principle, but
        # the number and order of the broadcasts below follow from them, so rank 0's map is THE map: one int64
        # broadcast, -1 = solved on its own
        table = torch.full((len(jobs),), -1, dtype=torch.int64)
        for index, rep_index in alias.items():
            table[index] = rep_index
        table = table.to(state.device)
        dist.broadcast(table, src=0)
        alias = {index: int(rep) for index, rep in enumerate(table.tolist()) if rep >= 0}
    solved = [index for index in range(len(jobs)) if index not in alias]   # dealt round-robin over the ranks
    owner_of = {index: position % world for position, index in enumerate(solved)}
    mine = [jobs[index] for index in solved if world == 1 or owner_of[index] == state.process_index]

    def solve(job, stream):
        module_name, cov_name, count_name, _vec, _val = job
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            work = covariance_factors[cov_name][module_name].to(device=state.device)
            noise = ops.STORAGE_NOISE.get(work.dtype, 0.0)   # a bf16-exported factor: the eigensolver's shift must clear its rounding
            if work.dtype not in (torch.float32, torch.float64):
                work = work.to(torch.float32)
            evals, evecs, _ = ops.eigh(work, float(covariance_factors[count_name][module_name].item()), noise_rel=noise)
            del work
            if release_covariances:
                del covariance_factors[cov_name][module_name]
            # cast on the solver's own stream: the fp64 results (2 d^2 x 8 bytes) do not pile up until the exchange below
            dtype = meta[(module_name, cov_name)][0]
            evals, evecs = evals.to(dtype), evecs.to(dtype).contiguous()
        return evals, evecs

    meta = {(job[0], job[1]): (covariance_factors[job[1]][job[0]].dtype, covariance_factors[job[1]][job[0]].shape[0]) for job in jobs}

    # The eigenproblems are independent and one Jacobi round kernel is latency / L2 bound far below the chip's
    # capacity, so several are kept in flight on separate HIP streams, each driven by its own host thread (the C
    # call releases the GIL and synchronises only its own stream, once per sweep).  Largest first.
    results = {}
    order = sorted(mine, key=lambda job: -meta[(job[0], job[1])][1])
    lanes = max(1, min(EIGH_STREAMS, len(order))) if state.device.type == "cuda" else 1
    if lanes == 1:
        for job in order:
            results[job[:2]] = solve(job, None)
    else:
        torch.cuda.synchronize(state.device)
        streams = [torch.cuda.Stream(device=state.device) for _ in range(lanes)]
        queue, lock = list(order), threading.Lock()

        def worker(stream):
            torch.cuda.set_device(state.device)
            while True:
                with lock:
                    if not queue:
                        return
                    job = queue.pop(0)
                results[job[:2]] = solve(job, stream)

        with ThreadPoolExecutor(max_workers=lanes) as pool:
            for future in [pool.submit(worker, stream) for stream in streams]:
                future.result()
        torch.cuda.synchronize(state.device)

    # Results travel in the factor's own dtype (fp32 unless the covariances were fp64): half the bytes of the fp64
    # solver output, tensor broadcasts over RCCL (no pickling), one per matrix.
    for index, (module_name, cov_name, count_name, vec_name, val_name) in enumerate(jobs):
        original_dtype, d = meta[(module_name, cov_name)]
        if release_covariances:
            covariance_factors[cov_name].pop(module_name, None)   # (problems solved on other ranks, shared solutions)
        if index in alias:   # shares the solution of an earlier matrix (already exchanged): its own copy, as a caller may modify either
            rep_name = jobs[alias[index]][0]
            out[val_name][module_name] = out[val_name][rep_name].clone()
            out[vec_name][module_name] = out[vec_name][rep_name].clone()
            continue
        owner = owner_of[index]
        if (module_name, cov_name) in results:
            evals, evecs = results.pop((module_name, cov_name))
        else:
            evals = torch.empty(d, dtype=original_dtype, device=state.device)
            evecs = torch.empty((d, d), dtype=original_dtype, device=state.device)
        if world > 1 or (state.use_distributed and dist.is_initialized()):   # (one forced rank: KF_DIST_FORCE)
            with exchange("eigen_broadcast", (evals.numel() + evecs.numel()) * evecs.element_size()):
                dist.broadcast(evals, src=owner)
                dist.broadcast(evecs, src=owner)
        target = "cpu" if cpu else state.device
        out[val_name][module_name] = evals.to(device=target).contiguous()
        out[vec_name][module_name] = evecs.to(device=target).contiguous()
    return out


def _fit_lambda_matrices_with_loader_impl(model: nn.Module, state: State, task: Task, loader: data.DataLoader,
                                    factor_args: FactorArguments, eigen_factors: Optional[FACTOR_TYPE] = None,
                                    tracked_module_names: Optional[List[str]] = None,
                                    disable_tqdm: bool = False, all_ranks: bool = False,
                                    cpu: bool = True) -> Tuple[torch.Tensor, FACTOR_TYPE]:
    """``all_ranks`` / ``cpu``: see ``fit_covariance_matrices_with_loader``."""
    del disable_tqdm
    update_factor_args(model, factor_args)
    if tracked_module_names is None:
        tracked_module_names = get_tracked_module_names(model)
    set_mode(model, ModuleMode.LAMBDA, tracked_module_names, release_memory=True)
    if eigen_factors is not None:
        for name in eigen_factors:
            set_factors(model, name, eigen_factors[name], clone=True, share=READ_ONLY_FACTORS)
    num_data_processed = torch.zeros((1,), dtype=torch.int64)
    enable_amp = factor_args.amp_dtype is not None
    scale = _loss_scale(factor_args)
    if scale != 1.0:
        set_gradient_scale(model, 1.0 / scale)
    set_side_stream(model, tracked_module_names, not factor_args.has_shared_parameters)   # see fit_covariance_matrices_with_loader
    try:
        for batch in loader:
            batch = send_to_device(batch, state.device)
            with no_sync(model, state):
                model.zero_grad(set_to_none=True)
                with autocast(device_type=state.device.type, enabled=enable_amp, dtype=factor_args.amp_dtype):
                    loss = task.compute_train_loss(batch=batch, model=model, sample=not factor_args.use_empirical_fisher)
                (loss * scale if scale != 1.0 else loss).backward()
            if factor_args.has_shared_parameters:
                finalize_iteration(model, tracked_module_names)
            num_data_processed.add_(find_batch_size(batch))
            del loss
    finally:
        set_side_stream(model, tracked_module_names, False)
    if state.use_distributed:
        synchronize_factors(model, LAMBDA_FACTOR_NAMES, tracked_module_names, state.device, extra=[num_data_processed])
    saved: FACTOR_TYPE = {}
    if state.is_main_process or all_ranks:
        for name in LAMBDA_FACTOR_NAMES:
            factor = load_factors(model, name, tracked_module_names, cpu=cpu,
                                  dtype=factor_args.lambda_dtype if name == LAMBDA_MATRIX_NAME else None)
            if len(factor) == 0:
                raise ValueError(f"Factor `{name}` has not been computed.")
            saved[name] = factor
    model.zero_grad(set_to_none=True)
    set_gradient_scale(model, 1.0)
    set_mode(model, ModuleMode.DEFAULT, release_memory=True)
    state.wait_for_everyone()
    return num_data_processed, saved


def fit_lambda_matrices_with_loader(*args, **kwargs) -> Tuple[torch.Tensor, FACTOR_TYPE]:
    """Stage entry point (signature of ``_fit_lambda_matrices_with_loader_impl``), cyclic GC paused."""
    with paused_gc():
        return _fit_lambda_matrices_with_loader_impl(*args, **kwargs)
