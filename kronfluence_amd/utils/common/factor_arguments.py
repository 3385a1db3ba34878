"""``FactorArguments`` presets (names and field values as the reference's
``utils/common/factor_arguments.py:6-63``, which its examples import).

On MI355X the dtype fields select the precision the kernels READ and EXPORT (bf16 fields route the
contractions to the bf16 MFMA engine); accumulation is always fp32.  The memory-saving switches
(``use_iterative_lambda_aggregation``, ``offload_activations_to_cpu``) are accepted for compatibility:
per-sample gradients are never materialised for Lambda and activations stay in the 288 GB of HBM.
"""

from __future__ import annotations

import torch

from kronfluence_amd.arguments import FactorArguments


def _preset(strategy: str, **fields) -> FactorArguments:
    return FactorArguments(strategy=strategy, **fields)


def _low_precision(dtype: torch.dtype, lambda_dtype: torch.dtype) -> dict:
    return dict(amp_dtype=dtype, activation_covariance_dtype=dtype, gradient_covariance_dtype=dtype,
                per_sample_gradient_dtype=dtype, lambda_dtype=lambda_dtype)


def default_factor_arguments(strategy: str = "ekfac") -> FactorArguments:
    return _preset(strategy)


def pytest_factor_arguments(strategy: str = "ekfac") -> FactorArguments:
    """Empirical Fisher (deterministic) and fp64 everywhere: the reference's test preset."""
    fp64 = torch.float64
    return _preset(strategy, use_empirical_fisher=True, activation_covariance_dtype=fp64, gradient_covariance_dtype=fp64,
                   per_sample_gradient_dtype=fp64, lambda_dtype=fp64)


def smart_low_precision_factor_arguments(strategy: str = "ekfac", dtype: torch.dtype = torch.bfloat16) -> FactorArguments:
    """Everything in ``dtype`` except Lambda, which stays fp32."""
    return _preset(strategy, **_low_precision(dtype, torch.float32))


def all_low_precision_factor_arguments(strategy: str = "ekfac", dtype: torch.dtype = torch.bfloat16) -> FactorArguments:
    return _preset(strategy, **_low_precision(dtype, dtype))


def reduce_memory_factor_arguments(strategy: str = "ekfac", dtype: torch.dtype = torch.bfloat16) -> FactorArguments:
    return _preset(strategy, use_iterative_lambda_aggregation=True, **_low_precision(dtype, dtype))


def extreme_reduce_memory_factor_arguments(strategy: str = "ekfac", module_partitions: int = 1,
                                           dtype: torch.dtype = torch.bfloat16) -> FactorArguments:
    return _preset(strategy, use_iterative_lambda_aggregation=True, offload_activations_to_cpu=True,
                   covariance_module_partitions=module_partitions, lambda_module_partitions=module_partitions,
                   **_low_precision(dtype, dtype))
