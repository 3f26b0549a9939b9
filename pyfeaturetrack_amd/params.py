"""KLT_TrackingContext -> klt_params (include/klt_gpu.h) and the three tap sets."""
from ._abi import KltAffineParams, KltParams
from .convolve import _computeKernels
from .klt_util import KLTComputeSmoothSigma

MAX_LEVELS = 8


def params_from_tc(tc):
    """Pack the numeric state of a tracking context.  Types follow how the reference consumes each
    value: C floats for the Cython `cdef float`s (trackFeaturesUtils.pyx:406-408), float32 for
    max_residue (compared against a numpy float32, trackFeatures.py:124), doubles elsewhere."""
    if tc.lighting_insensitive:
        # trackFeaturesUtils.pyx:434-435 raises the same exception inside the Newton loop
        raise Exception("Not implemented")
    if tc.window_width != tc.window_height:
        # the reference's patch loops are transposed for non-square windows
        raise ValueError("window_width must equal window_height")
    if tc.nPyramidLevels > MAX_LEVELS or tc.nPyramidLevels < 1:
        raise ValueError("nPyramidLevels must be in 1..%d" % MAX_LEVELS)
    p = KltParams()
    p.mindist = int(tc.mindist)
    p.window_width = int(tc.window_width)
    p.window_height = int(tc.window_height)
    p.smoothBeforeSelecting = int(bool(tc.smoothBeforeSelecting))
    p.retainTrackers = int(bool(tc.retainTrackers))
    p.nSkippedPixels = int(tc.nSkippedPixels)
    p.max_iterations = int(tc.max_iterations)
    p.nPyramidLevels = int(tc.nPyramidLevels)
    p.subsampling = int(tc.subsampling)
    p.use_max_residue = int(tc.max_residue is not None)
    p.min_determinant = float(tc.min_determinant)
    p.min_displacement = float(tc.min_displacement)
    p.step_factor = float(tc.step_factor)
    p.max_residue = float(tc.max_residue) if tc.max_residue is not None else 0.0
    p.min_eigenvalue = float(tc.min_eigenvalue)
    p.grad_sigma = float(tc.grad_sigma)
    p.smooth_sigma = float(KLTComputeSmoothSigma(tc))
    p.pyramid_sigma = float(tc.pyramid_sigma_fact * tc.subsampling)
    p.borderx = float(tc.borderx)
    p.bordery = float(tc.bordery)
    return p


def taps_from_params(p):
    """[(gauss, deriv)] for smoothing, pyramid and gradient sigma (klt_set_kernels `which` 0, 1, 2)."""
    return [_computeKernels(p.smooth_sigma), _computeKernels(p.pyramid_sigma), _computeKernels(p.grad_sigma)]


def affine_params_from_tc(tc):
    """klt.py:67-73 -> klt_affine_params (mode -1 = consistency check off)."""
    a = KltAffineParams()
    a.mode = int(tc.affineConsistencyCheck)
    a.window_width = int(tc.affine_window_width)
    a.window_height = int(tc.affine_window_height)
    a.max_iterations = int(tc.affine_max_iterations)
    a.max_residue = float(tc.affine_max_residue)
    a.min_displacement = float(tc.affine_min_displacement)
    a.max_displacement_differ = float(tc.affine_max_displacement_differ)
    return a
