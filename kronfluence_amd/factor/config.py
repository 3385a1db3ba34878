"""Strategy registry (``FactorConfig``) and the EK-FAC strategy on MI355X.

Interface as the reference's ``factor/config.py:30-125`` (seven ``requires_*`` properties,
``prepare``, ``precondition_gradient``, registry ``FactorConfig.CONFIGS``).  ``ekfac`` is the strategy
the north star names; ``identity`` / ``diagonal`` / ``kfac`` reuse its kernels:
K-FAC is EK-FAC with ``Lambda = lambda_G (x) lambda_A``, the diagonal strategy is the Lambda kernel and the
``mul`` GEMM epilogue without rotations.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Optional

import torch

from kronfluence_amd import ops
from kronfluence_amd.utils.constants import (
    ACTIVATION_EIGENVALUES_NAME,
    ACTIVATION_EIGENVECTORS_NAME,
    GRADIENT_EIGENVALUES_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    LAMBDA_MATRIX_NAME,
    NUM_LAMBDA_PROCESSED,
)

STORAGE_TYPE = Dict[str, Any]


class FactorStrategy:
    IDENTITY = "identity"
    DIAGONAL = "diagonal"
    KFAC = "kfac"
    EKFAC = "ekfac"


class UnknownFactorStrategyError(KeyError, NotImplementedError):
    """An unregistered strategy name: a ``KeyError`` like the reference's plain dict lookup (``FactorConfig.CONFIGS[name]``), and a
    ``NotImplementedError`` for callers that ask whether a strategy exists on this engine."""

    def __str__(self) -> str:   # KeyError would print the repr of the message
        return str(self.args[0]) if self.args else ""


class _Registry(dict):
    def __missing__(self, key: str) -> "FactorConfig":
        known = ", ".join(sorted(self))
        raise UnknownFactorStrategyError(
            f"Factor strategy `{key}` is not part of the MI355X hot path (available: {known}). "
            "See SURVEY.md section 8(f) for the widening order."
        )


class FactorConfig(ABC):
    """Describes which factors a strategy needs and how it preconditions a per-sample gradient."""

    CONFIGS: Dict[str, "FactorConfig"] = _Registry()

    def __init_subclass__(cls, factor_strategy: Optional[str] = None, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)
        if factor_strategy is not None:
            FactorConfig.CONFIGS[factor_strategy] = cls()

    @property
    @abstractmethod
    def requires_covariance_matrices(self) -> bool: ...

    @property
    @abstractmethod
    def requires_eigendecomposition(self) -> bool: ...

    @property
    @abstractmethod
    def requires_lambda_matrices(self) -> bool: ...

    @property
    @abstractmethod
    def requires_eigendecomposition_for_lambda(self) -> bool: ...

    @property
    @abstractmethod
    def requires_covariance_matrices_for_precondition(self) -> bool: ...

    @property
    @abstractmethod
    def requires_eigendecomposition_for_precondition(self) -> bool: ...

    @property
    @abstractmethod
    def requires_lambda_matrices_for_precondition(self) -> bool: ...

    def prepare(self, storage: STORAGE_TYPE, score_args: Any, device: torch.device) -> None:
        """One-time transformation of the stored factors before scoring."""

    @abstractmethod
    def precondition_gradient(self, gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
        """``gradient``: ``[batch, out, in]`` -> preconditioned gradient of the same shape."""


def _rotate_scale_rotate(gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
    """Generic form on a materialised ``[b, O, I']`` gradient (reference ``config.py:341-353``):
    ``Qg ((Qg^T g Qa) o Lambda^-1) Qa^T`` as four batched MFMA GEMMs; the elementwise product is
    fused into the second one's epilogue.  The trackers use the factored ``kf_precondition``
    instead whenever the gradient's factors are available."""
    q_a, q_g = storage[ACTIVATION_EIGENVECTORS_NAME], storage[GRADIENT_EIGENVECTORS_NAME]
    lam_inv = storage[LAMBDA_MATRIX_NAME]
    g = gradient.contiguous()
    b, o, ip = g.shape
    dev = g.device
    q_a, q_g, lam_inv = (t.to(device=dev, dtype=torch.float32).contiguous() for t in (q_a, q_g, lam_inv))
    t1 = torch.empty((b * o, ip), dtype=torch.float32, device=dev)
    ops.gemm(t1, ip, 0, ops.view(g, 0, ip, 1, b * o, ip), ops.view(q_a, 0, 1, ip, ip, ip))  # g Qa
    t2 = torch.empty((b, o, ip), dtype=torch.float32, device=dev)
    ops.gemm(t2, ip, o * ip, ops.view(q_g, 0, 1, o, o, o), ops.view(t1, o * ip, 1, ip, ip, o), batch=b,
             mul=lam_inv)  # (Qg^T .) o Lambda^-1
    ops.gemm(t1, ip, 0, ops.view(t2, 0, ip, 1, b * o, ip), ops.view(q_a, 0, ip, 1, ip, ip))  # . Qa^T
    ops.gemm(t2, ip, o * ip, ops.view(q_g, 0, o, 1, o, o), ops.view(t1, o * ip, 1, ip, ip, o), batch=b)  # Qg .
    return t2


class Identity(FactorConfig, factor_strategy=FactorStrategy.IDENTITY):
    """No preconditioning (reference ``factor/config.py:128-166``): scores are plain gradient dot products."""

    requires_covariance_matrices = False
    requires_eigendecomposition = False
    requires_eigendecomposition_for_lambda = False
    requires_lambda_matrices = False
    requires_covariance_matrices_for_precondition = False
    requires_eigendecomposition_for_precondition = False
    requires_lambda_matrices_for_precondition = False

    def precondition_gradient(self, gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
        del storage
        return gradient


class Diagonal(FactorConfig, factor_strategy=FactorStrategy.DIAGONAL):
    """Diagonal Fisher (reference ``factor/config.py:169-222``): ``Lambda = sum_b g_b^2`` in parameter space."""

    requires_covariance_matrices = False
    requires_eigendecomposition = False
    requires_eigendecomposition_for_lambda = False
    requires_lambda_matrices = True
    requires_covariance_matrices_for_precondition = False
    requires_eigendecomposition_for_precondition = False
    requires_lambda_matrices_for_precondition = True

    def prepare(self, storage: STORAGE_TYPE, score_args: Any, device: torch.device) -> None:
        n_lambda = float(storage[NUM_LAMBDA_PROCESSED].item())
        lam = storage[LAMBDA_MATRIX_NAME].to(device=device, dtype=torch.float32)
        storage[LAMBDA_MATRIX_NAME] = ops.inv_lambda(lam, n_lambda, score_args.damping_factor)
        storage[NUM_LAMBDA_PROCESSED] = None

    @torch.no_grad()
    def precondition_gradient(self, gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
        """``g o Lambda^-1`` on a materialised gradient (kf_mul_bcast)."""
        return ops.mul_bcast(gradient, storage[LAMBDA_MATRIX_NAME].to(device=gradient.device, dtype=torch.float32))


class Kfac(FactorConfig, factor_strategy=FactorStrategy.KFAC):
    """K-FAC (Martens & Grosse, 2015; reference ``factor/config.py:225-285``): the EK-FAC preconditioner with
    ``Lambda[o, i] = lambda_G[o] * lambda_A[i]`` taken from the eigenvalues instead of a fitted correction."""

    requires_covariance_matrices = True
    requires_eigendecomposition = True
    requires_eigendecomposition_for_lambda = False
    requires_lambda_matrices = False
    requires_covariance_matrices_for_precondition = False
    requires_eigendecomposition_for_precondition = True
    requires_lambda_matrices_for_precondition = False

    def prepare(self, storage: STORAGE_TYPE, score_args: Any, device: torch.device) -> None:
        for name in (ACTIVATION_EIGENVECTORS_NAME, GRADIENT_EIGENVECTORS_NAME):
            storage[name] = storage[name].to(device=device, dtype=torch.float32).contiguous()
        lam_a = storage[ACTIVATION_EIGENVALUES_NAME].to(device=device, dtype=torch.float32).contiguous()
        lam_g = storage[GRADIENT_EIGENVALUES_NAME].to(device=device, dtype=torch.float32).contiguous()
        o, ip = lam_g.numel(), lam_a.numel()
        lam = torch.empty((o, ip), dtype=torch.float32, device=device)
        ops.gemm(lam, ip, 0, ops.view(lam_g, 0, 1, 1, o, 1), ops.view(lam_a, 0, 1, 1, ip, 1))  # outer product, depth 1
        storage[LAMBDA_MATRIX_NAME] = ops.inv_lambda(lam, 1.0, score_args.damping_factor)
        storage[NUM_LAMBDA_PROCESSED] = None
        storage[ACTIVATION_EIGENVALUES_NAME] = None
        storage[GRADIENT_EIGENVALUES_NAME] = None

    @torch.no_grad()
    def precondition_gradient(self, gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
        return _rotate_scale_rotate(gradient, storage)


class Ekfac(FactorConfig, factor_strategy=FactorStrategy.EKFAC):
    """Eigenvalue-corrected K-FAC (George et al., 2018)."""

    requires_covariance_matrices = True
    requires_eigendecomposition = True
    requires_eigendecomposition_for_lambda = True
    requires_lambda_matrices = True
    requires_covariance_matrices_for_precondition = False
    requires_eigendecomposition_for_precondition = True
    requires_lambda_matrices_for_precondition = True

    def prepare(self, storage: STORAGE_TYPE, score_args: Any, device: torch.device) -> None:
        """Reference ``factor/config.py:322-339``: eigenvectors to the precondition dtype, Lambda
        replaced by ``1 / (Lambda / n + damping)`` (fp64 arithmetic, kf_inv_lambda).  Everything
        stays RESIDENT in HBM -- the reference parks the result on the CPU and re-uploads it on
        every query batch (config.py:347-349)."""
        # (bf16 preconditioner + bf16-stored eigenvectors: the bf16 call chain, kf_precondition_bf16, reads no fp32 copy; the paths
        #  that do convert on first use, ``BaseTracker._eigenvectors32`` -- at Llama-3-8B's depth the fp32 copies are 99 GB)
        lazy = (getattr(score_args, "precondition_dtype", None) == torch.bfloat16 and getattr(score_args, "score_dtype", None) == torch.bfloat16)
        for name in (ACTIVATION_EIGENVECTORS_NAME, GRADIENT_EIGENVECTORS_NAME):
            if lazy and storage[name].dtype == torch.bfloat16:
                storage[name] = storage[name].to(device=device).contiguous()
            else:
                storage[name] = storage[name].to(device=device, dtype=torch.float32).contiguous()
        storage[ACTIVATION_EIGENVALUES_NAME] = None
        storage[GRADIENT_EIGENVALUES_NAME] = None
        n_lambda = float(storage[NUM_LAMBDA_PROCESSED].item())
        lam = storage[LAMBDA_MATRIX_NAME].to(device=device, dtype=torch.float32)
        storage[LAMBDA_MATRIX_NAME] = ops.inv_lambda(lam, n_lambda, score_args.damping_factor)
        storage[NUM_LAMBDA_PROCESSED] = None

    @torch.no_grad()
    def precondition_gradient(self, gradient: torch.Tensor, storage: STORAGE_TYPE) -> torch.Tensor:
        return _rotate_scale_rotate(gradient, storage)
