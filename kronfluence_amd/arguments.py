"""``FactorArguments`` / ``ScoreArguments``: field names, defaults and validation follow the
reference's ``kronfluence/arguments.py:38-274`` (they are part of the public API and are what
``factor_arguments.json`` / ``score_arguments.json`` serialise).

Dtype fields select the dtype results are EXPORTED in.  On MI355X every accumulation happens in
fp32 (fp64 for the eigensolver and the Lambda reciprocal) irrespective of these fields, which is at
least the reference's precision.  Fields that only choose an intermediate precision or a memory
strategy of the reference's implementation are accepted, serialised and honoured as follows:

    eigendecomposition_dtype           the eigensolver is fp64 (kf_eigh_f64) whatever is asked: the reference's default, and at
                                       least as accurate as its float32 option; results are exported in the factor's dtype
    per_sample_gradient_dtype          per-sample gradients are formed from the hooked tensors as they arrive (bf16 under
                                       autocast, else fp32) with fp32 accumulation; they are rounded to bf16 only where the
                                       reference's bf16 presets round them too (bf16 ``score_dtype`` / ``lambda_dtype``)
    query_gradient_svd_dtype           the low-rank factorisation (range finder + kf_eigh_small_batched) works in fp32 / fp64
    use_iterative_lambda_aggregation   honoured where a batch of rotated per-sample gradients would otherwise be materialised
                                       (post-processed / shared-parameter gradients, LambdaTracker._update_from_gradient);
                                       the factored paths never materialise them
    offload_activations_to_cpu         honoured (BaseTracker._cache_activation): the hooked input waits for its gradient in
                                       host memory; FactorArguments' flag for the Lambda stage, ScoreArguments' for the score stages
"""

from dataclasses import asdict, dataclass, fields
from typing import Any, Dict, Optional

import torch


@dataclass
class Arguments:
    def to_dict(self) -> Dict[str, Any]:
        out = asdict(self)
        for key, value in out.items():
            if isinstance(value, torch.dtype):
                out[key] = str(value)
        return out

    def to_str_dict(self) -> Dict[str, str]:
        return {key: str(value) for key, value in self.to_dict().items()}


@dataclass
class FactorArguments(Arguments):
    strategy: str = "ekfac"
    use_empirical_fisher: bool = False
    amp_dtype: Optional[torch.dtype] = None
    amp_scale: float = 2.0**16
    has_shared_parameters: bool = False

    covariance_max_examples: Optional[int] = 100_000
    covariance_data_partitions: int = 1
    covariance_module_partitions: int = 1
    activation_covariance_dtype: torch.dtype = torch.float32
    gradient_covariance_dtype: torch.dtype = torch.float32

    eigendecomposition_dtype: torch.dtype = torch.float64

    lambda_max_examples: Optional[int] = 100_000
    lambda_data_partitions: int = 1
    lambda_module_partitions: int = 1
    use_iterative_lambda_aggregation: bool = False
    offload_activations_to_cpu: bool = False
    per_sample_gradient_dtype: torch.dtype = torch.float32
    lambda_dtype: torch.dtype = torch.float32

    def __post_init__(self) -> None:
        for name in ("covariance_max_examples", "lambda_max_examples"):
            value = getattr(self, name)
            if value is not None and value <= 0:
                raise ValueError(f"`{name}` must be `None` or positive.")
        partitions = (self.covariance_data_partitions, self.covariance_module_partitions,
                      self.lambda_data_partitions, self.lambda_module_partitions)
        if min(partitions) <= 0:
            raise ValueError("All data and module partitions must be positive.")


@dataclass
class ScoreArguments(Arguments):
    damping_factor: Optional[float] = 1e-08
    amp_dtype: Optional[torch.dtype] = None
    offload_activations_to_cpu: bool = False

    data_partitions: int = 1
    module_partitions: int = 1

    compute_per_module_scores: bool = False
    compute_per_token_scores: bool = False

    query_gradient_accumulation_steps: int = 1
    query_gradient_low_rank: Optional[int] = None
    use_full_svd: bool = False
    aggregate_query_gradients: bool = False
    aggregate_train_gradients: bool = False

    use_measurement_for_self_influence: bool = False

    query_gradient_svd_dtype: torch.dtype = torch.float32
    per_sample_gradient_dtype: torch.dtype = torch.float32
    precondition_dtype: torch.dtype = torch.float32
    score_dtype: torch.dtype = torch.float32

    def __post_init__(self) -> None:
        if self.damping_factor is not None and self.damping_factor < 0:
            raise ValueError("`damping_factor` must be `None` or positive.")
        if min(self.data_partitions, self.module_partitions) <= 0:
            raise ValueError("Both data and module partitions must be positive.")
        if self.query_gradient_accumulation_steps <= 0:
            raise ValueError("`query_gradient_accumulation_steps` must be positive.")
        if self.query_gradient_low_rank is not None and self.query_gradient_low_rank <= 0:
            raise ValueError("`query_gradient_low_rank` must be `None` or positive.")


def unsupported_score_options(score_args: ScoreArguments) -> Dict[str, Any]:
    """Options outside the accelerated hot path ("next" rows); the score stage
    rejects them explicitly instead of silently computing something else."""
    del score_args   # every ScoreArguments option is implemented (ranks above 88 leave the in-LDS eigensolver, see ops.eigh_small)
    return {}


def all_field_names(cls) -> set:
    return {f.name for f in fields(cls)}
