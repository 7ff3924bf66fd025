"""Clustering surface: KCenters / KHybrid / KMedoids (reference
enspara/cluster/__init__.py)."""
from . import hybrid  # noqa: F401
from . import kcenters  # noqa: F401
from . import kmedoids  # noqa: F401
from . import util  # noqa: F401
from .hybrid import KHybrid  # noqa: F401
from .kcenters import KCenters  # noqa: F401
from .kmedoids import KMedoids  # noqa: F401
