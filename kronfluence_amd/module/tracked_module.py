"""``TrackedModule``: the wrapper installed around every tracked ``nn.Linear`` / ``nn.Conv2d``.

This is the drop-in boundary of the hot path (reference
``module/tracked_module.py:49-416``): same constructor, same ``ModuleMode`` table, same ``storage``
keys, same model-facing behaviour (the wrapped forward is untouched; a zero parameter makes the
output require grad so a tensor hook fires although every weight is frozen).

Two groups of per-module operators exist:

* the reference's operator API (``get_flattened_activation`` ... ``compute_pairwise_score``), kept
  so that custom subclasses and tests written against the reference keep working; and
* fused accumulate-style operators (``accumulate_*``) that the trackers call on the hot path: they
  hand the hooked activation / output-gradient tensors straight to the HIP kernels (no flatten,
  ``cat``, cast or per-sample-gradient materialisation in between).
"""

from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Tuple, Type, Union

import torch
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.arguments import FactorArguments, ScoreArguments
from kronfluence_amd.factor.config import FactorConfig
from kronfluence_amd.module.tracker.base import BaseTracker
from kronfluence_amd.module.tracker.factor import CovarianceTracker, LambdaTracker
from kronfluence_amd.module.tracker.gradient import GradientTracker
from kronfluence_amd.module.tracker.pairwise_score import PairwiseScoreTracker
from kronfluence_amd.module.tracker.precondition import PreconditionTracker
from kronfluence_amd.module.tracker.self_score import SelfScoreTracker, SelfScoreWithMeasurementTracker
from kronfluence_amd.utils.constants import (
    ACCUMULATED_PRECONDITIONED_GRADIENT_NAME,
    AGGREGATED_GRADIENT_NAME,
    COVARIANCE_FACTOR_NAMES,
    EIGENDECOMPOSITION_FACTOR_NAMES,
    LAMBDA_FACTOR_NAMES,
    PAIRWISE_SCORE_MATRIX_NAME,
    PRECONDITIONED_GRADIENT_NAME,
    SELF_SCORE_VECTOR_NAME,
)


class ModuleMode(str):
    """String-valued mode constants (the reference uses a ``str`` enum with the same values)."""

    DEFAULT = "default"
    COVARIANCE = "covariance"
    LAMBDA = "lambda"
    PRECONDITION_GRADIENT = "precondition_gradient"
    PAIRWISE_SCORE = "pairwise_score"
    SELF_SCORE = "self_score"
    SELF_MEASUREMENT_SCORE = "self_measurement_score"
    GRADIENT_AGGREGATION = "gradient_aggregation"


class TrackedModule(nn.Module):
    SUPPORTED_MODULES: Dict[Type[nn.Module], Any] = {}

    def __init_subclass__(cls, module_type: Optional[Type[nn.Module]] = None, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)
        if module_type is not None:
            cls.SUPPORTED_MODULES[module_type] = cls

    def __init__(self, name: str, original_module: nn.Module, factor_args: Optional[FactorArguments] = None,
                 score_args: Optional[ScoreArguments] = None,
                 per_sample_gradient_process_fnc: Optional[Callable] = None) -> None:
        super().__init__()
        self.name = name
        self.original_module = original_module
        # requires_grad zero: keeps autograd alive through a fully frozen layer
        self._constant = nn.Parameter(torch.zeros(1, dtype=original_module.weight.dtype, requires_grad=True))
        self.current_mode = ModuleMode.DEFAULT
        self.factor_args = factor_args if factor_args is not None else FactorArguments()
        self.score_args = score_args if score_args is not None else ScoreArguments()
        self.per_sample_gradient_process_fnc = per_sample_gradient_process_fnc
        self._trackers: Dict[str, BaseTracker] = {
            ModuleMode.DEFAULT: BaseTracker(self),
            ModuleMode.COVARIANCE: CovarianceTracker(self),
            ModuleMode.LAMBDA: LambdaTracker(self),
            ModuleMode.PRECONDITION_GRADIENT: PreconditionTracker(self),
            ModuleMode.PAIRWISE_SCORE: PairwiseScoreTracker(self),
            ModuleMode.GRADIENT_AGGREGATION: GradientTracker(self),
            ModuleMode.SELF_SCORE: SelfScoreTracker(self),
            ModuleMode.SELF_MEASUREMENT_SCORE: SelfScoreWithMeasurementTracker(self),
        }
        self.attention_mask: Optional[torch.Tensor] = None
        self.gradient_scale: float = 1.0
        self.einsum_path: Optional[List[int]] = None  # kept for attribute compatibility; unused
        # (scores buffer [Q, N], column offset) shared by all layers during the train pass
        self.score_sink: Optional[Tuple[torch.Tensor, int]] = None
        # True while storage["preconditioned_gradient"] holds eigenbasis-resident queries (PreconditionTracker)
        self.queries_in_eigenbasis: bool = False
        # > 0 while the held preconditioned query gradients carry that many trailing zero columns (bf16 engine, odd I')
        self.query_padding: int = 0
        # how many query gradients the score stage is about to hold in this module (None: unknown) -- lets the
        # PreconditionTracker make one allocation per layer (QueryBuffer)
        self.query_capacity: Optional[int] = None
        # True only while the pairwise query loop of a multi-rank job is running: the PreconditionTracker may then issue the
        # all-gather of a freshly preconditioned block from the backward hook (the loop calls ``synchronize`` after every query
        # batch).  Every other user of PRECONDITION_GRADIENT mode (self-influence with measurement, ...) exchanges nothing.
        self.async_query_gather: bool = False
        # True only inside the stage loops of this package: the trackers may launch their hooks' kernels on a second stream
        # (BaseTracker._run_beside) -- the loops join it before the results are read.  Direct users of the module API read
        # ``storage`` right after backward(): everything stays on their stream.
        self.side_stream_ok: bool = False
        self.storage: Dict[str, Any] = {}
        for key in (COVARIANCE_FACTOR_NAMES + EIGENDECOMPOSITION_FACTOR_NAMES + LAMBDA_FACTOR_NAMES
                    + [AGGREGATED_GRADIENT_NAME, PRECONDITIONED_GRADIENT_NAME,
                       ACCUMULATED_PRECONDITIONED_GRADIENT_NAME, PAIRWISE_SCORE_MATRIX_NAME, SELF_SCORE_VECTOR_NAME]):
            self.storage[key] = None

    # -- model-facing -------------------------------------------------------------------------------
    def forward(self, inputs: torch.Tensor, *args: Any, **kwargs: Any) -> torch.Tensor:
        outputs = self.original_module(inputs, *args, **kwargs)
        if outputs.requires_grad:
            return outputs
        # keep the layer's own output dtype: adding the fp32 parameter to a bf16 (autocast) output would promote the
        # whole downstream block -- and the next tracked layer's hooked activation -- to fp32
        return outputs + self._constant.to(dtype=outputs.dtype)

    # -- bookkeeping (reference tracked_module.py:170-318) -----------------------------------------
    def prepare_storage(self, device: torch.device) -> None:
        FactorConfig.CONFIGS[self.factor_args.strategy].prepare(storage=self.storage, score_args=self.score_args,
                                                                device=device)

    def update_factor_args(self, factor_args: FactorArguments) -> None:
        self.factor_args = factor_args

    def update_score_args(self, score_args: ScoreArguments) -> None:
        self.score_args = score_args

    def get_factor(self, factor_name: str) -> Optional[torch.Tensor]:
        return self.storage.get(factor_name)

    def release_factor(self, factor_name: str) -> None:
        if self.storage.get(factor_name) is not None:
            self.storage[factor_name] = None

    def set_factor(self, factor_name: str, factor: Any) -> None:
        if factor_name in self.storage:
            self.storage[factor_name] = factor

    def set_mode(self, mode: str, release_memory: bool = False) -> None:
        self._trackers[self.current_mode].release_hooks()
        self.einsum_path = None
        self.current_mode = mode
        if release_memory:
            for tracker in self._trackers.values():
                tracker.release_memory()
        self._trackers[self.current_mode].register_hooks()

    def set_attention_mask(self, attention_mask: Optional[torch.Tensor] = None) -> None:
        self.attention_mask = attention_mask

    def set_gradient_scale(self, scale: float = 1.0) -> None:
        self.gradient_scale = scale

    def finalize_iteration(self) -> None:
        self._trackers[self.current_mode].finalize_iteration()

    def exist(self) -> bool:
        return self._trackers[self.current_mode].exist()

    def synchronize(self, num_processes: int) -> None:
        self._trackers[self.current_mode].synchronize(num_processes=num_processes)

    def truncate(self, keep_size: int) -> None:
        self._trackers[self.current_mode].truncate(keep_size=keep_size)

    def accumulate_iterations(self) -> None:
        self._trackers[self.current_mode].accumulate_iterations()

    def finalize_all_iterations(self) -> None:
        self._trackers[self.current_mode].finalize_all_iterations()

    # -- reference operator API (abstract in tracked_module.py:320-416) -----------------------------
    def get_flattened_activation(self, input_activation: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        raise NotImplementedError

    def get_flattened_gradient(self, output_gradient: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, int]]:
        raise NotImplementedError

    def compute_per_sample_gradient(self, input_activation: torch.Tensor, output_gradient: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def compute_pairwise_score(self, preconditioned_gradient: torch.Tensor, input_activation: torch.Tensor,
                               output_gradient: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def compute_summed_gradient(self, input_activation: torch.Tensor, output_gradient: torch.Tensor) -> torch.Tensor:
        """``einsum("b...i,b...o->io")[None]`` of reference ``linear.py:56-66`` / ``conv2d.py:134-162`` -> ``[1, O, I']``."""
        g, a, ones = self.gradient_factors(input_activation, output_gradient)
        out = torch.zeros((1, g.shape[-1], a.shape[-1] + int(ones)), dtype=torch.float32, device=g.device)
        self.accumulate_summed_gradient(out, g, a, ones, 1.0)
        return out

    @staticmethod
    def accumulate_summed_gradient(total: torch.Tensor, g: torch.Tensor, a: torch.Tensor, ones: bool, scale: float) -> None:
        """``total[0] += scale * G^T [A, 1]`` over all ``b * R`` rows: one GEMM, depth ``b R``."""
        g, a = g.contiguous(), a.contiguous()
        rows = g.shape[0] * g.shape[1]
        o, i = g.shape[-1], a.shape[-1]
        ip = i + int(ones)
        ops.gemm(total, ip, 0, ops.view(g, 0, 1, o, o, rows), ops.view(a, 0, 1, i, i, rows, ones_row=ones),
                 alpha=scale, beta=1.0)

    def compute_self_measurement_score(self, preconditioned_gradient: torch.Tensor, input_activation: torch.Tensor,
                                       output_gradient: torch.Tensor) -> torch.Tensor:
        """``einsum("bio,b...i,b...o->b")`` of reference ``linear.py:124-138`` / ``conv2d.py:211-227``: the per-sample
        gradient on the MFMA engine, then one ``kf_rowwise_dot``."""
        g, a, ones = self.gradient_factors(input_activation, output_gradient)
        psg = ops.per_sample_gradient(g, a, ones)
        p = preconditioned_gradient.contiguous()
        if p.dtype not in (torch.float32, torch.bfloat16):
            p = p.to(torch.float32)
        scores = torch.zeros(psg.shape[0], dtype=torch.float32, device=psg.device)
        ops.rowwise_dot(scores, p, psg, None, accumulate=False)
        return scores

    # -- fused hot-path operators ------------------------------------------------------------------
    @property
    def has_bias(self) -> bool:
        return self.original_module.bias is not None

    def accumulate_activation_covariance(self, cov: Optional[torch.Tensor], count: Optional[torch.Tensor],
                                         input_activation: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """``cov += X'^T X'``, ``count += rows`` for the hooked input; allocates on first use."""
        raise NotImplementedError

    def accumulate_gradient_covariance(self, cov: Optional[torch.Tensor], count: Optional[torch.Tensor],
                                       output_gradient: torch.Tensor, alpha: float) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def gradient_factors(self, input_activation: torch.Tensor, output_gradient: torch.Tensor
                         ) -> Tuple[torch.Tensor, torch.Tensor, bool]:
        """``(G [b,R,O], A [b,R,I], append_ones)`` with ``g_b = sum_r G[b,r,:]^T [A[b,r,:], 1]``."""
        raise NotImplementedError
