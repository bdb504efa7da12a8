"""Drop-in for the reference's `lib.metrics.StructuralLosses` package (built from
lib/metrics/pytorch_structural_losses by its Makefile:73-77)."""
from .nn_distance import nn_distance  # noqa: F401
from .match_cost import match_cost  # noqa: F401
