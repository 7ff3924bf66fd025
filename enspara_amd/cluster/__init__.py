"""Clustering surface: KCenters / KHybrid / KMedoids (reference
enspara/cluster/__init__.py)."""
from . import kcenters  # noqa: F401
from . import util  # noqa: F401
from .kcenters import KCenters  # noqa: F401
