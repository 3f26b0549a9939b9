"""Gaussian / Gaussian-derivative taps and the separable convolutions built on them.

Reference: convolve.py.  The taps are generated on the host in FP64 exactly as
`_computeKernels` does (convolve.py:27-93: exp(-i^2/2s^2) over 71 candidates, tails below 1 %
of the peak dropped, gauss normalised to sum 1, derivative normalised so that sum(-i*d) = 1)
and handed to the device as FP64 constants.  The convolutions themselves
(`_convolveSeparate`, convolve.py:208-214, i.e. scipy.ndimage.convolve1d along axis 1 then
axis 0, FP64 accumulate, f32 store after each pass, `reflect` borders) run as HIP kernels.

Deviation: the reference re-uses the previously computed taps when the new sigma
is within 0.05 of the last one; here taps are always exact for the sigma requested.
"""
import math

import numpy as np

from .error import KLTError

MAX_KERNEL_WIDTH = 71
_TAIL_FACTOR = 0.01
_cache = {}


class ConvolutionKernel:
    """Name kept from the reference (convolve.py:14-17); holds `width` and `data`."""
    def __init__(self, maxKernelWidth=MAX_KERNEL_WIDTH):
        self.width = None
        self.data = [0.0] * maxKernelWidth


def _trimmed_width(values, peak):
    width = len(values)
    k = 0
    while abs(values[k] / peak) < _TAIL_FACTOR:
        k += 1
        width -= 2
    return width


def _computeKernels(sigma):
    """(gauss taps, derivative taps) as Python float lists -- convolve.py:27-93."""
    key = float(sigma)
    hit = _cache.get(key)
    if hit is not None:
        return list(hit[0]), list(hit[1])
    assert sigma >= 0.0
    half = MAX_KERNEL_WIDTH // 2
    two_var = 2 * sigma * sigma
    g_full = [math.exp(-i * i / two_var) for i in range(-half, half + 1)]
    d_full = [-i * g for i, g in zip(range(-half, half + 1), g_full)]
    gw = _trimmed_width(g_full, 1.0)
    dw = _trimmed_width(d_full, float(sigma * math.exp(-0.5)))
    if gw == MAX_KERNEL_WIDTH or dw == MAX_KERNEL_WIDTH:
        KLTError("(_computeKernels) maxKernelWidth {0} is too small for a sigma of {1}".format(MAX_KERNEL_WIDTH, sigma))
    g0 = (MAX_KERNEL_WIDTH - gw) // 2
    d0 = (MAX_KERNEL_WIDTH - dw) // 2
    gauss = g_full[g0:g0 + gw]
    deriv = d_full[d0:d0 + dw]
    den = 0.0
    for v in gauss:
        den += v
    gauss = [v / den for v in gauss]
    dh = dw // 2
    den = 0.0
    for i in range(-dh, dh + 1):
        den -= i * deriv[i + dh]
    deriv = [v / den for v in deriv]
    _cache[key] = (tuple(gauss), tuple(deriv))
    return gauss, deriv


def KLTGetKernelWidths(sigma):
    g, d = _computeKernels(sigma)
    return len(g), len(d)


def _as_f32(img):
    a = np.ascontiguousarray(img, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("expected a 2-D float image")
    return a


def _convolveSeparate(imgin, horiz_kernel, vert_kernel):
    """convolve.py:208-219: the image convolved along its rows with `horiz_kernel`, then along its columns with `vert_kernel` -- the
    reference's one general separable convolution (SciPy branch: scipy.ndimage.convolve1d, axis 1 then axis 0, `reflect` borders, FP64
    accumulation, the image's f32 after each pass).  Any two tap sequences of 1 to 71 numbers (the reference's `_computeKernels` never
    makes more), odd or even counts, symmetric or not.  f32 image in and out (what the reference passes: `img.convert("F")` arrays;
    another dtype is converted first -- scipy would keep it).  Runs on the GPU (klt_convolve_separate_f32)."""
    from .backend import default_context
    h = [float(v) for v in np.asarray(horiz_kernel, dtype=np.float64).ravel()]
    v = [float(x) for x in np.asarray(vert_kernel, dtype=np.float64).ravel()]
    if not (1 <= len(h) <= MAX_KERNEL_WIDTH and 1 <= len(v) <= MAX_KERNEL_WIDTH):
        raise ValueError("(_convolveSeparate) 1 to {0} taps per direction".format(MAX_KERNEL_WIDTH))
    ctx = default_context()
    with ctx.lock:
        return ctx.convolve_separate(_as_f32(imgin), h, v)


def KLTComputeSmoothedImage(img, sigma):
    """f32 image -> f32 image smoothed with (gauss, gauss) -- convolve.py:254-264.  Runs on the GPU."""
    from .backend import default_context
    g, _ = _computeKernels(sigma)
    ctx = default_context()
    with ctx.lock:
        return ctx.smooth(_as_f32(img), g)


def KLTComputeGradients(img, sigma):
    """(gradx, grady) -- convolve.py:226-248.  Runs on the GPU."""
    from .backend import default_context
    g, d = _computeKernels(sigma)
    ctx = default_context()
    with ctx.lock:
        return ctx.gradients(_as_f32(img), g, d)
