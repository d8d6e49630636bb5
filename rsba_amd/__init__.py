"""rsba_amd — MI355X-native rolling-shutter bundle-adjustment hot path (drop-in for rsba's Ceres path).

Host-side Python here is a thin driver over the C-ABI in include/rsba_amd.h (librsba_amd.so, built
from rsba_amd/csrc by __graft_entry__.build()).  All arithmetic on the path runs in the HIP kernels;
there is no CPU fallback — a missing extension raises.
"""
from .problem import BAProblem, GLOBAL, HORIZONTAL, VERTICAL, apply_gauge_masks  # noqa: F401
