"""mrs_optic_flow_amd -- MI355X-native core for the mrs_optic_flow hot path.

The product is ``libmof_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/mof.h``); this package holds its sources (``csrc/``), the ctypes binding and
thin Python handles named after the reference's processors. Importing the package is
cheap and does not need a GPU; constructing any processor loads the library and fails
loudly if it is missing or no HIP device is present -- there is no CPU fallback.
"""
from . import _capi  # noqa: F401
from ._capi import MofError, MofLibraryError  # noqa: F401
from .engine import BlockMethod, FastSpacedBMMethod, FftMethod, ScaleRotationEstimator, pinned_empty, release_captured  # noqa: F401

__all__ = ["FftMethod", "BlockMethod", "FastSpacedBMMethod", "ScaleRotationEstimator", "MofError", "MofLibraryError",
           "release_captured", "pinned_empty"]
