"""Exception types of the public API (same names as the reference's ``utils/exceptions.py``)."""


class FactorsNotFoundError(ValueError):
    """Factors required by a stage (covariances, eigenvectors, Lambda) are missing."""


class TrackedModuleNotFoundError(ValueError):
    """The model contains no (or not the requested) ``TrackedModule``."""


class IllegalTaskConfigurationError(ValueError):
    """The ``Task`` names modules that do not exist or nothing in the model can be tracked."""


class UnsupportableModuleError(NotImplementedError):
    """The module's configuration (e.g. asymmetric string padding) cannot be handled."""
