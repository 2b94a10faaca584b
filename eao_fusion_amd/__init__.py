"""eao_fusion_amd -- MI355X-native ORB front-end + local-BA hot path of EAO-Fusion.

Host-side mirror (Python flavour, for tests and benchmarks) of the reference's operator classes; the C++
flavour that drops into Tracking.cc / LocalMapping.cc lives in include/eaofusion/.  All compute happens in
libeaofusion_hip.so (hand-written HIP kernels for gfx950) through the C-ABI in include/eao_fusion.h.
"""
from ._lib import EaoError, load  # noqa: F401
from .orb import KP_DTYPE, ORBextractor, compute_stereo_matches  # noqa: F401
from .matcher import ORBmatcher, distinctive_descriptors, hamming_best2, hamming_matrix  # noqa: F401
from .optimizer import Optimizer  # noqa: F401
