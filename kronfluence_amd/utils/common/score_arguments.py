"""``ScoreArguments`` presets (names and field values as the reference's
``utils/common/score_arguments.py:8-85``)."""

from __future__ import annotations

from typing import Optional

import torch

from kronfluence_amd.arguments import ScoreArguments


def default_score_arguments(damping_factor: Optional[float] = 1e-08,
                            query_gradient_low_rank: Optional[int] = None) -> ScoreArguments:
    """Defaults; with low-rank query gradients ten query batches are held per train pass (they are small)."""
    steps = 10 if query_gradient_low_rank is not None else 1
    return ScoreArguments(damping_factor=damping_factor, query_gradient_low_rank=query_gradient_low_rank,
                          query_gradient_accumulation_steps=steps)


def pytest_score_arguments(damping_factor: Optional[float] = 1e-08,
                           query_gradient_low_rank: Optional[int] = None) -> ScoreArguments:
    fp64 = torch.float64
    return ScoreArguments(damping_factor=damping_factor, query_gradient_low_rank=query_gradient_low_rank,
                          query_gradient_svd_dtype=fp64, score_dtype=fp64, per_sample_gradient_dtype=fp64,
                          precondition_dtype=fp64)


def _low_precision(args: ScoreArguments, dtype: torch.dtype, precondition_dtype: torch.dtype) -> ScoreArguments:
    args.amp_dtype = dtype
    args.score_dtype = dtype
    args.per_sample_gradient_dtype = dtype
    args.precondition_dtype = precondition_dtype
    args.query_gradient_svd_dtype = torch.float32
    return args


def smart_low_precision_score_arguments(damping_factor: Optional[float] = 1e-08,
                                        query_gradient_low_rank: Optional[int] = None,
                                        dtype: torch.dtype = torch.bfloat16) -> ScoreArguments:
    """Gradients and scores in ``dtype``, preconditioning in fp32."""
    return _low_precision(default_score_arguments(damping_factor, query_gradient_low_rank), dtype, torch.float32)


def all_low_precision_score_arguments(damping_factor: Optional[float] = 1e-08,
                                      query_gradient_low_rank: Optional[int] = None,
                                      dtype: torch.dtype = torch.bfloat16) -> ScoreArguments:
    return _low_precision(default_score_arguments(damping_factor, query_gradient_low_rank), dtype, dtype)


def reduce_memory_score_arguments(damping_factor: Optional[float] = 1e-08, query_gradient_low_rank: Optional[int] = None,
                                  dtype: torch.dtype = torch.bfloat16) -> ScoreArguments:
    args = all_low_precision_score_arguments(damping_factor, query_gradient_low_rank, dtype)
    args.offload_activations_to_cpu = True
    return args


def extreme_reduce_memory_score_arguments(damping_factor: Optional[float] = 1e-08, module_partitions: int = 4,
                                          query_gradient_low_rank: Optional[int] = None,
                                          dtype: torch.dtype = torch.bfloat16) -> ScoreArguments:
    args = reduce_memory_score_arguments(damping_factor, query_gradient_low_rank, dtype)
    args.module_partitions = module_partitions
    return args
