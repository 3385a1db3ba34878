"""Storage keys and on-disk names.

These strings are part of the drop-in contract -- they are the keys of ``TrackedModule.storage`` and the stems of
the safetensors files, so factors and scores written by this engine and by the reference are interchangeable
(reference ``utils/constants.py:13-79``; values pinned by tests/test_host_logic.py).  The names are generated from
the two Kronecker sides ("activation" = layer input, "gradient" = pre-activation pseudo-gradient) rather than listed.
"""

from typing import Dict, List, Optional, Tuple, Union

import torch

# -- type aliases ------------------------------------------------------------------------------------
FACTOR_TYPE = Dict[str, Dict[str, torch.Tensor]]          # {factor name: {module name: tensor}}
SCORE_TYPE = Dict[str, torch.Tensor]                      # {"all_modules" | module name: scores}
PARTITION_TYPE = Tuple[int, int]                          # (data partition, module partition)
PRECONDITIONED_GRADIENT_TYPE = Optional[Union[torch.Tensor, List[torch.Tensor]]]  # dense, or [left, right] factors

# -- numerics ----------------------------------------------------------------------------------------
LAMBDA_DTYPE = torch.float64            # dtype of the Lambda reciprocal (factor/config.py:331-338)
HEURISTIC_DAMPING_SCALE = 0.1           # damping_factor=None -> 0.1 * mean(Lambda / n)
DISTRIBUTED_SYNC_INTERVAL = 1_000

# -- directory / file stems ----------------------------------------------------------------------------
FACTOR_SAVE_PREFIX, SCORE_SAVE_PREFIX = "factors_", "scores_"
FACTOR_ARGUMENTS_NAME, SCORE_ARGUMENTS_NAME = "factor", "score"
ALL_MODULE_NAME = "all_modules"

# -- stage 1 and 2a: one covariance / eigenbasis per Kronecker side ------------------------------------
_SIDES = ("activation", "gradient")
ACTIVATION_COVARIANCE_MATRIX_NAME, GRADIENT_COVARIANCE_MATRIX_NAME = (f"{side}_covariance" for side in _SIDES)
NUM_ACTIVATION_COVARIANCE_PROCESSED, NUM_GRADIENT_COVARIANCE_PROCESSED = (
    f"num_{side}_covariance_processed" for side in _SIDES)
ACTIVATION_EIGENVECTORS_NAME, GRADIENT_EIGENVECTORS_NAME = (f"{side}_eigenvectors" for side in _SIDES)
ACTIVATION_EIGENVALUES_NAME, GRADIENT_EIGENVALUES_NAME = (f"{side}_eigenvalues" for side in _SIDES)

COVARIANCE_FACTOR_NAMES = [f"{side}_covariance" for side in _SIDES] + [f"num_{side}_covariance_processed" for side in _SIDES]
EIGENDECOMPOSITION_FACTOR_NAMES = [f"{side}_eigen{what}" for side in _SIDES for what in ("vectors", "values")]

# -- stage 2b ---------------------------------------------------------------------------------------------
LAMBDA_MATRIX_NAME, NUM_LAMBDA_PROCESSED = "lambda_matrix", "num_lambda_processed"
LAMBDA_FACTOR_NAMES = [LAMBDA_MATRIX_NAME, NUM_LAMBDA_PROCESSED]

# -- stage 3: transient per-module state ----------------------------------------------------------------------
PRECONDITIONED_GRADIENT_NAME = "preconditioned_gradient"
ACCUMULATED_PRECONDITIONED_GRADIENT_NAME = f"accumulated_{PRECONDITIONED_GRADIENT_NAME}"
AGGREGATED_GRADIENT_NAME = "aggregated_gradient"
PAIRWISE_SCORE_MATRIX_NAME, SELF_SCORE_VECTOR_NAME = "pairwise_score_matrix", "self_score_vector"
