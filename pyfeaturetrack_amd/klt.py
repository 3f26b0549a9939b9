"""Public KLT types: tracking context, feature record, status codes.

Reference: klt.py.  Same names, same attribute set and the same derived values *as the
reference computes them under Python 3* (true division: a 7x7 window gives a half-width of
3.5 and a default border of 30.0, not the 24 of the C library -- SURVEY.md A.1).  The
numeric state lives in a `klt_params` POD (include/klt_gpu.h) pushed across the C ABI.
"""
from __future__ import print_function

import math
import time

import numpy as np

from .convolve import KLTGetKernelWidths
from .error import KLTError, KLTWarning  # noqa: F401  (re-exported like the reference's star imports)
from .klt_util import KLTComputeSmoothSigma

# example1.py:52 uses time.clock(), removed in Python 3.8; give it back so that the reference's
# own driver script runs unchanged on top of this backend.
if not hasattr(time, "clock"):
    time.clock = time.perf_counter


class kltState:
    """klt.py:23-29"""
    KLT_TRACKED = 0
    KLT_NOT_FOUND = -1
    KLT_SMALL_DET = -2
    KLT_MAX_ITERATIONS = -3
    KLT_OOB = -4
    KLT_LARGE_RESIDUE = -5


def _pyramidSigma(tc):
    """klt.py:196-197"""
    return tc.pyramid_sigma_fact * tc.subsampling


class KLT_TrackingContext:
    """klt.py:42-190.  Plain attributes; nothing is validated until a KLT* call uses them."""

    def __init__(self):
        self.mindist = 10
        self.window_width = 7
        self.window_height = 7
        self.sequentialMode = False
        self.retainTrackers = False
        self.smoothBeforeSelecting = True
        self.writeInternalImages = False
        self.lighting_insensitive = False
        self.min_eigenvalue = 1
        self.min_determinant = 0.01
        self.max_iterations = 10
        self.min_displacement = 0.1
        self.max_residue = None
        self.grad_sigma = 1.0
        self.smooth_sigma_fact = 0.1
        self.pyramid_sigma_fact = 0.9
        self.step_factor = 1.0
        self.nSkippedPixels = 0
        self.pyramid_last = None
        self.pyramid_last_gradx = None
        self.pyramid_last_grady = None
        # affine consistency check (klt.py:67-73).  Not implemented by the reference.
        self.affineConsistencyCheck = -1
        self.affine_window_width = 15
        self.affine_window_height = 15
        self.affine_max_iterations = 10
        self.affine_max_residue = 10.0
        self.affine_min_displacement = 0.02
        self.affine_max_displacement_differ = 1.5

        self.KLTChangeTCPyramid(15)
        self.KLTUpdateTCBorder()

    # -- window sanity, shared by every entry point of the reference (klt.py:87-101, :143-157)
    def _check_window(self, who):
        for name, what in (("window_width", "width"), ("window_height", "height")):
            v = getattr(self, name)
            if v % 2 != 1:
                v += 1
                setattr(self, name, v)
                KLTWarning("({0}) Window {1} must be odd.  Changing to {2}.\n".format(who, what, v))
        for name, what in (("window_width", "width"), ("window_height", "height")):
            if getattr(self, name) < 3:
                setattr(self, name, 3)
                KLTWarning("({0}) Window {1} must be at least three.  \nChanging to 3.\n".format(who, what))

    def KLTChangeTCPyramid(self, search_range):
        """Choose nPyramidLevels / subsampling for a search range -- klt.py:84-128."""
        self._check_window("KLTChangeTCPyramid")
        half = min(self.window_width, self.window_height) / 2.0
        ratio = float(search_range) / half
        if ratio < 1.0:
            self.nPyramidLevels = 1
        elif ratio <= 9.0:
            self.nPyramidLevels = 2
            self.subsampling = 2 if ratio <= 3.0 else (4 if ratio <= 5.0 else 8)
        else:
            # search_range = half * (8^L - 1) / 7, rounded up
            self.nPyramidLevels = int(math.log(7.0 * ratio + 1.0) / math.log(8.0) + 0.99)
            self.subsampling = 8

    def KLTUpdateTCBorder(self):
        """Border lost to convolution and windows -- klt.py:137-189 (Python-3 float halves)."""
        self._check_window("KLTUpdateTCBorder")
        levels = self.nPyramidLevels
        ss = self.subsampling
        window_half = max(self.window_width, self.window_height) / 2
        smooth_half = KLTGetKernelWidths(KLTComputeSmoothSigma(self))[0] / 2
        pyramid_half = KLTGetKernelWidths(_pyramidSigma(self))[0] / 2
        invalid = smooth_half
        for _ in range(1, levels):
            invalid = int((float(invalid) + pyramid_half) / ss + 0.99)
        border = (invalid + window_half) * ss ** (levels - 1)
        self.borderx = border
        self.bordery = border


class KLT_Feature:
    """klt.py:249-263.  The reference's __init__ assigns locals only; real attributes are set on
    first placement (selectGoodFeatures.py:117-128).  Here they always exist: x, y, val are set by __init__, the
    affine-consistency fields read as their defaults until something assigns them (a list of 20 000 features is
    created and updated in Python on every selection, so the constructor stays small)."""

    __slots__ = ("x", "y", "val", "aff_img", "aff_img_gradx", "aff_img_grady",
                 "aff_x", "aff_y", "aff_Axx", "aff_Ayx", "aff_Axy", "aff_Ayy", "__weakref__")
    _AFF_DEFAULTS = {"aff_img": None, "aff_img_gradx": None, "aff_img_grady": None, "aff_x": -1.0, "aff_y": -1.0,
                     "aff_Axx": 1.0, "aff_Ayx": 0.0, "aff_Axy": 0.0, "aff_Ayy": 1.0}

    def __init__(self):
        self.x = -1
        self.y = -1
        self.val = kltState.KLT_NOT_FOUND

    def __getattr__(self, name):                  # only reached for a slot that was never assigned
        try:
            return KLT_Feature._AFF_DEFAULTS[name]
        except KeyError:
            raise AttributeError(name)

    def _reset_affine(self):
        """Back to the state of a newly placed feature (selectGoodFeatures.py:120-128)."""
        for name in KLT_Feature._AFF_DEFAULTS:
            try:
                delattr(self, name)
            except AttributeError:
                pass


_REC_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("val", np.int32), ("aux", np.int32)])   # == klt_feat


def _lost_records(shape):
    rec = np.zeros(shape, _REC_DTYPE)
    rec["x"] = -1
    rec["y"] = -1
    rec["val"] = kltState.KLT_NOT_FOUND
    return rec


class KLT_FeatureHistory:
    """One feature over all frames (klt.py:272-276 is an empty stub; upstream KLT 1.3.4's KLT_FeatureHistoryRec).
    `rec` is the [nFrames] record array (x, y, val); fh[frame] gives a KLT_Feature copy."""

    def __init__(self, nFrames=0):
        self.nFrames = nFrames
        self.rec = _lost_records(nFrames)

    def __len__(self):
        return self.nFrames

    def __getitem__(self, frame):
        return _feature_of(self.rec[frame])


class KLT_FeatureTable:
    """All features over all frames (klt.py:278-283 is an empty stub; upstream KLT 1.3.4's KLT_FeatureTableRec).
    `rec` is a [nFrames, nFeatures] array of 16-byte records (x, y, val, aux) -- exactly the layout of the
    device-side table KLTTrackSequence fills, so a tracked sequence comes back with ONE download.  Upstream indexes
    feature-major (ft->feature[feat][frame]); `ft.feature(feat, frame)` mirrors that."""

    def __init__(self, nFrames=0, nFeatures=0):
        self.nFrames, self.nFeatures = nFrames, nFeatures
        self.rec = _lost_records((nFrames, nFeatures))

    x = property(lambda self: self.rec["x"])
    y = property(lambda self: self.rec["y"])
    val = property(lambda self: self.rec["val"])

    def feature(self, feat, frame):
        return _feature_of(self.rec[frame, feat])


def _feature_of(r):
    f = KLT_Feature()
    f.x, f.y, f.val = float(r["x"]), float(r["y"]), int(r["val"])
    return f


def KLTPrintTrackingContext(tc):
    """klt.py:285-313 -- same lines, same order."""
    print(tc)
    print("\n\nTracking context:\n")
    for name in ("mindist", "window_width", "window_height", "sequentialMode", "smoothBeforeSelecting",
                 "writeInternalImages"):
        print("\t{0} = {1}".format(name, getattr(tc, name)))
    for name in ("min_eigenvalue", "min_determinant", "min_displacement", "max_iterations", "max_residue",
                 "grad_sigma", "smooth_sigma_fact", "pyramid_sigma_fact", "nSkippedPixels", "borderx", "bordery",
                 "nPyramidLevels", "subsampling"):
        print("\t{0} = {1}".format(name, getattr(tc, name)))
    print("\n\tpyramid_last = {0}".format(tc.pyramid_last))
    print("\tpyramid_last_gradx = {0}".format(tc.pyramid_last_gradx))
    print("\tpyramid_last_grady = {0}".format(tc.pyramid_last_grady))
    print("\n")


def KLTCountRemainingFeatures(fl):
    """klt.py:319-325"""
    return sum(1 for feat in fl if feat.val >= 0)
