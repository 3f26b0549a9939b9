"""KLTPyramid (reference: pyramid.py:14-77).

The tracker does not use this class -- it builds all three pyramids of a frame on the device in
one enqueue (klt_build_pyramids).  The class is kept for API compatibility: level 0 is the input,
level i is level i-1 smoothed with sigma = subsampling * sigma_fact and sampled at
(ss*y + ss/2, ss*x + ss/2), dimensions int(n / ss) per level -- all levels in one call on the GPU
(klt_pyramid_f32: only the surviving columns / rows are evaluated, one download at the end).
"""
import numpy as np

from .convolve import _computeKernels
from .error import KLTError

_ALLOWED = (2, 4, 8, 16, 32)


class KLTPyramid:
    def __init__(self, ncols, nrows, subsampling, nlevels):
        if subsampling not in _ALLOWED:
            KLTError("(_KLTCreatePyramid)  Pyramid's subsampling must be either 2, 4, 8, 16, or 32")
        self.subsampling = subsampling
        self.nLevels = nlevels
        self.img = [None] * nlevels
        self.ncols = []
        self.nrows = []
        for _ in range(nlevels):
            self.ncols.append(ncols)
            self.nrows.append(nrows)
            ncols /= subsampling       # true division, as in the reference (levels >= 1 hold floats)
            nrows /= subsampling

    def Compute(self, img, sigma_fact):
        from .backend import default_context
        img = np.ascontiguousarray(img, np.float32)
        assert self.ncols[0] == img.shape[1] and self.nrows[0] == img.shape[0]
        self.img[0] = img
        if self.nLevels > 1:
            gauss, _ = _computeKernels(self.subsampling * sigma_fact)
            ctx = default_context()
            with ctx.lock:
                self.img[1:] = ctx.pyramid(img, self.subsampling, self.nLevels, gauss)
