"""Storage keys and on-disk names.  These strings are part of the drop-in contract: they are the keys
of ``TrackedModule.storage`` and the safetensors file names, identical to the reference's
``utils/constants.py:13-79`` so that factors and scores are interchangeable between the two."""

from typing import Dict, List, Optional, Tuple, Union

import torch

FACTOR_TYPE = Dict[str, Dict[str, torch.Tensor]]
PARTITION_TYPE = Tuple[int, int]
SCORE_TYPE = Dict[str, torch.Tensor]
PRECONDITIONED_GRADIENT_TYPE = Optional[Union[torch.Tensor, List[torch.Tensor]]]

FACTOR_SAVE_PREFIX = "factors_"
SCORE_SAVE_PREFIX = "scores_"
FACTOR_ARGUMENTS_NAME = "factor"
SCORE_ARGUMENTS_NAME = "score"

DISTRIBUTED_SYNC_INTERVAL = 1_000
HEURISTIC_DAMPING_SCALE = 0.1

ACTIVATION_COVARIANCE_MATRIX_NAME = "activation_covariance"
GRADIENT_COVARIANCE_MATRIX_NAME = "gradient_covariance"
NUM_ACTIVATION_COVARIANCE_PROCESSED = "num_activation_covariance_processed"
NUM_GRADIENT_COVARIANCE_PROCESSED = "num_gradient_covariance_processed"
COVARIANCE_FACTOR_NAMES = [
    ACTIVATION_COVARIANCE_MATRIX_NAME,
    GRADIENT_COVARIANCE_MATRIX_NAME,
    NUM_ACTIVATION_COVARIANCE_PROCESSED,
    NUM_GRADIENT_COVARIANCE_PROCESSED,
]

ACTIVATION_EIGENVECTORS_NAME = "activation_eigenvectors"
ACTIVATION_EIGENVALUES_NAME = "activation_eigenvalues"
GRADIENT_EIGENVECTORS_NAME = "gradient_eigenvectors"
GRADIENT_EIGENVALUES_NAME = "gradient_eigenvalues"
EIGENDECOMPOSITION_FACTOR_NAMES = [
    ACTIVATION_EIGENVECTORS_NAME,
    ACTIVATION_EIGENVALUES_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    GRADIENT_EIGENVALUES_NAME,
]

LAMBDA_MATRIX_NAME = "lambda_matrix"
NUM_LAMBDA_PROCESSED = "num_lambda_processed"
LAMBDA_FACTOR_NAMES = [LAMBDA_MATRIX_NAME, NUM_LAMBDA_PROCESSED]

PRECONDITIONED_GRADIENT_NAME = "preconditioned_gradient"
ACCUMULATED_PRECONDITIONED_GRADIENT_NAME = "accumulated_preconditioned_gradient"
AGGREGATED_GRADIENT_NAME = "aggregated_gradient"
PAIRWISE_SCORE_MATRIX_NAME = "pairwise_score_matrix"
SELF_SCORE_VECTOR_NAME = "self_score_vector"

ALL_MODULE_NAME = "all_modules"
LAMBDA_DTYPE = torch.float64
