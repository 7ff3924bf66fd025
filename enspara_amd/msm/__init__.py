"""MSM construction on the device: transition counts, row normalisation,
leading eigenpairs (reference enspara/msm)."""
from . import builders  # noqa: F401
from . import transition_matrices  # noqa: F401
from .transition_matrices import assigns_to_counts, eigenspectrum, eq_probs  # noqa: F401
from .msm import MSM  # noqa: F401
from .timescales import implied_timescales  # noqa: F401
from .trimming import TrimMapping, trim_disconnected  # noqa: F401
