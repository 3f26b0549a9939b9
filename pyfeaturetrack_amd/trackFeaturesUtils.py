"""The reference's Cython module trackFeaturesUtils (setup.py:9) on the MI355X backend: the two functions the tracker calls per
feature and level (trackFeatures.py:102-106), with the reference's names, arguments and results.

KLTTrackFeatures does not come through here -- klt_track runs all features and levels in one launch.  This module is the reference's
literal native boundary for callers (and parity checks) that use it directly; every call uploads the planes it is given.
computeIntensityDifference (the residue test of _trackFeature calls it, trackFeatures.py:120) and computeGradientSum
(trackFeaturesUtils.pyx:90-97, :130-142) are here as well: the bilinear samples come from the device (klt_extract_patch_f32), the
f32 subtraction behind them is numpy's.  The SciPy-optimiser helpers minFunc / jacobian (:342-388) belong to an alternative path the
tracker does not take and are not provided.
"""
import ctypes as C

import numpy as np

from .backend import default_context


def _plane(a, what):
    a = np.ascontiguousarray(a, np.float32)
    if a.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got {0}): {1}".format(a.ndim, what))
    return a


def extractImagePatchSlow(img, x, y, height, width):
    """trackFeaturesUtils.pyx:14-51: float32 [height, width] bilinear samples of `img` around (x, y) (both truncated to f32 first,
    as the Cython signature does).  Raises AssertionError when the footprint leaves the image, like the reference (:35)."""
    img = _plane(img, "img")
    height, width = int(height), int(width)
    patch = np.empty((height, width), np.float32)
    ctx = default_context()
    with ctx.lock:
        rc = ctx._lib.klt_extract_patch_f32(ctx._h, img.ctypes.data, img.shape[1], img.shape[0], float(np.float32(x)),
                                            float(np.float32(y)), width, height, patch.ctypes.data)
        if rc == -1 and b"leaves the image" in (ctx._lib.klt_last_error(ctx._h) or b""):
            raise AssertionError("ix - hw >= 0 and iy - hh >= 0 and ix + hw + 2 <= ncols and iy + hh + 2 <= nrows")
        ctx._check(rc)
    return patch


def _f32_2d(a, what):
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2:
        raise ValueError("Buffer dtype mismatch or wrong number of dimensions (expected a 2-D float32 array): {0}".format(what))
    return a


def computeIntensityDifference(img1Patch, img2, x2, y2, workingPatch, out):
    """trackFeaturesUtils.pyx:61-97: `workingPatch` (its shape is the window) receives the bilinear samples of `img2` around (x2, y2),
    `out` (flat float32, window size) the differences img1Patch - workingPatch in row-major order.  Returns None."""
    work = _f32_2d(workingPatch, "workingPatch")
    p1 = _f32_2d(img1Patch, "img1Patch")
    if not isinstance(out, np.ndarray) or out.dtype != np.float32 or out.ndim != 1:
        raise ValueError("Buffer dtype mismatch or wrong number of dimensions (expected a 1-D float32 array): out")
    h, w = work.shape
    work[...] = extractImagePatchSlow(img2, x2, y2, h, w)
    hh, hw = h // 2, w // 2                                   # (the loops run over 2 * half + 1 rows / columns, :80-81)
    rows, cols = 2 * hh + 1, 2 * hw + 1
    out[:rows * cols] = (p1[:rows, :cols] - work[:rows, :cols]).ravel()
    return None


def computeGradientSum(img1GradxPatch, gradx2, x2, y2, workingPatch, out, row):
    """trackFeaturesUtils.pyx:107-142: `workingPatch` receives the bilinear samples of `gradx2` around (x2, y2); column `row` of the
    2-D float32 `out` receives -img1GradxPatch - workingPatch, entry (j, i) at index j * workingPatch.shape[0] + i (the reference's
    indexing: row-major for the square windows the tracker uses)."""
    work = _f32_2d(workingPatch, "workingPatch")
    p1 = _f32_2d(img1GradxPatch, "img1GradxPatch")
    out = _f32_2d(out, "out")
    h, w = work.shape
    work[...] = extractImagePatchSlow(gradx2, x2, y2, h, w)
    s = -p1[:h, :w] - work
    if h == w:
        out[:h * w, row] = s.ravel()
    else:
        for j in range(h):
            out[j * h:j * h + w, row] = s[j]
    return None


def trackFeatureIterateCKLT(x2, y2, img1GradxPatch, img1GradyPatch, img1Patch, img2, gradx2, grady2, tc):
    """trackFeaturesUtils.pyx:393-459: (x2, y2, status, iteration) after the Newton loop of one feature at one level."""
    if getattr(tc, "lighting_insensitive", False):
        raise Exception("Not implemented")                    # :437-438
    gxp, gyp, ip = _plane(img1GradxPatch, "img1GradxPatch"), _plane(img1GradyPatch, "img1GradyPatch"), _plane(img1Patch, "img1Patch")
    i2, gx2, gy2 = _plane(img2, "img2"), _plane(gradx2, "gradx2"), _plane(grady2, "grady2")
    width, height = int(tc.window_width), int(tc.window_height)
    if not (gxp.shape == gyp.shape == ip.shape == (height, width)) or not (i2.shape == gx2.shape == gy2.shape):
        raise ValueError("patches must be window_height x window_width, and the three images of frame 2 equal in shape")
    ctx = default_context()
    xo, yo, st, it = C.c_float(), C.c_float(), C.c_int(), C.c_int()
    with ctx.lock:
        ctx._check(ctx._lib.klt_track_iterate_f32(ctx._h, float(np.float32(x2)), float(np.float32(y2)), gxp.ctypes.data, gyp.ctypes.data,
                                                 ip.ctypes.data, width, height, i2.ctypes.data, gx2.ctypes.data, gy2.ctypes.data,
                                                 i2.shape[1], i2.shape[0], float(tc.step_factor), float(tc.min_determinant),
                                                 float(tc.min_displacement), int(tc.max_iterations),
                                                 C.byref(xo), C.byref(yo), C.byref(st), C.byref(it)))
    return xo.value, yo.value, st.value, it.value


__all__ = ["extractImagePatchSlow", "trackFeatureIterateCKLT", "computeIntensityDifference", "computeGradientSum"]
