"""Geometry helpers on the hot path: point-vs-set distances in feature space
(reference enspara/geometry/libdist.pyx)."""
from . import libdist  # noqa: F401
