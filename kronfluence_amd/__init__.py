"""MI355X-native EK-FAC influence engine behind kronfluence's Analyzer / Task / prepare_model API.

>>> from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, Task, prepare_model
"""

__version__ = "0.1.0"

from kronfluence_amd.arguments import FactorArguments, ScoreArguments  # noqa: E402,F401
from kronfluence_amd.task import Task  # noqa: E402,F401


def __getattr__(name):
    # Analyzer / prepare_model pull in the HIP bindings; import them lazily so that host-only
    # utilities (arguments, samplers, constants) stay importable everywhere.
    if name in ("Analyzer", "prepare_model"):
        from kronfluence_amd import analyzer

        return getattr(analyzer, name)
    if name == "utils":   # the reference exports its ``utils`` package at the top level
        import importlib

        return importlib.import_module("kronfluence_amd.utils")
    raise AttributeError(name)
