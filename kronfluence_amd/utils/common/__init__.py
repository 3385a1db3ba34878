"""Argument presets with the reference's names (``kronfluence/utils/common``)."""
