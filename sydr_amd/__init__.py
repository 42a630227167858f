"""sydr_amd -- MI355X-native GNSS correlator engine behind SyDR's Channel/Receiver plugin surface.

Hot path only: PCPS acquisition, E/P/L tracking correlators and PRN replica generation run as
hand-written HIP kernels (sydr_amd/csrc) behind the C-ABI in include/sydr_amd.h.
"""
from ._lib import LIB_PATH, SdrError, device_count, load  # noqa: F401

__version__ = "0.1.0"
