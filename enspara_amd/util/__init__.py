"""Host-side helpers around the clustering path (reference enspara/util)."""
from . import load  # noqa: F401
