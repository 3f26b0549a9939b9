"""pyfeaturetrack_amd -- MI355X-native KLT feature tracker with PyFeatureTrack's API surface.

Only the hot path of TimSC/PyFeatureTrack is here: pyramid build, min-eigenvalue corner selection,
per-feature Newton tracking -- as hand-written HIP kernels (csrc/) behind a C ABI
(include/klt_gpu.h) bound with ctypes.  Importing the package does not touch the GPU; the first
KLT* call opens the device and fails loudly if libkltgpu.so or the GPU is missing.
"""
from .klt import (KLT_Feature, KLT_FeatureHistory, KLT_FeatureTable, KLT_TrackingContext,  # noqa: F401
                  KLTCountRemainingFeatures, KLTPrintTrackingContext, kltState)

__version__ = "0.1.0"


def __getattr__(name):
    # lazy: these modules bind the HIP library on first use
    import importlib
    for mod in ("selectGoodFeatures", "trackFeatures", "writeFeatures", "storeFeatures", "trackSequence"):
        m = importlib.import_module("." + mod, __name__)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
