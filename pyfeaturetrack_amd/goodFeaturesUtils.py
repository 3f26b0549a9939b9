"""The reference's Cython module goodFeaturesUtils (setup.py:8) on the MI355X backend: same function name, arguments and results.

KLTSelectGoodFeatures does not come through here -- klt_select keeps the scores on the device and hands back only the selected
records.  This module is the reference's literal native boundary for callers (and parity checks) that use it directly.
"""
import ctypes as C

import numpy as np

from .backend import default_context


def ScanImageForGoodFeatures(gradxArr, gradyArr, borderx, bordery, window_hw, window_hh, nSkippedPixels):
    """goodFeaturesUtils.pyx:35-73: (pointlistx, pointlisty, pointlistval) of every candidate window, y outer / x inner.
    The border and window arguments are truncated to C ints as Cython does (30.0 -> 30, 3.5 -> 3)."""
    gx = np.ascontiguousarray(gradxArr, np.float32)
    gy = np.ascontiguousarray(gradyArr, np.float32)
    if gx.ndim != 2 or gx.shape != gy.shape:
        raise ValueError("Buffer has wrong number of dimensions (expected 2) or the gradient images differ in shape")
    nrows, ncols = gx.shape
    bx, by, hw, hh, skip = int(borderx), int(bordery), int(window_hw), int(window_hh), int(nSkippedPixels)
    step = skip + 1
    xs = np.arange(bx, ncols - bx, step, dtype=np.int32)
    ys = np.arange(by, nrows - by, step, dtype=np.int32)
    val = np.empty(max(1, len(xs) * len(ys)), np.float32)
    ctx = default_context()
    nx, ny = C.c_int(), C.c_int()
    with ctx.lock:
        ctx._check(ctx._lib.klt_scan_good_features_f32(ctx._h, gx.ctypes.data, gy.ctypes.data, ncols, nrows, bx, by, hw, hh, skip,
                                                      val.ctypes.data, val.size, C.byref(nx), C.byref(ny)))
    assert (nx.value, ny.value) == (len(xs), len(ys))
    n = nx.value * ny.value
    pointlistx = list(np.tile(xs, ny.value))                 # numpy int32 scalars, as `pointlistx.extend(xRow)` leaves them (:67-71)
    pointlisty = list(np.repeat(ys, nx.value))
    return pointlistx, pointlisty, val[:n].tolist()


__all__ = ["ScanImageForGoodFeatures"]
