"""MI355X-native k-centers / RMSD clustering hot path with enspara's surface.

    from enspara_amd.cluster import KCenters, KHybrid
    from enspara_amd import ra

The arithmetic runs in hand-written HIP kernels for gfx950
(enspara_amd/csrc, C ABI in include/enspara_hip.h).  There is no CPU
fallback: without the built library and a HIP device the 'rmsd' path raises.
"""
from . import exception  # noqa: F401

__version__ = "0.1.0"
