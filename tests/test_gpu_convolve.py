"""a-3 by name: `_convolveSeparate(imgin, horiz_kernel, vert_kernel)` (convolve.py:208-219) on the device -- klt_convolve_separate_f32 over
the generic hconv / vconv kernels -- against vectors the reference itself produced (tests/golden/gen_convolve_separate.py) and against
the oracle on random shapes and tap lists.  Every image is compared bit for bit."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "convolve_separate.npz"))


def _same(a, b, what):
    assert a.dtype == b.dtype == np.float32 and a.shape == b.shape, what
    bad = np.flatnonzero(a.ravel() != b.ravel())
    assert bad.size == 0, "%s: %d of %d samples differ, first at %d: %r vs %r" % (what, bad.size, a.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]])


def test_convolve_separate_by_the_reference_name_vs_reference_vectors(golden, img0):
    """the Python entry point under the reference's name, all sixteen golden cases: the reference's own tap pairs (symmetric /
    antisymmetric branches of correlate1d) and the ones only `_convolveSeparate` can be handed (the general branch, even counts, a
    single tap, taps wider than the image, symmetry within / beyond DBL_EPSILON)"""
    from pyfeaturetrack_amd import convolve
    from pyfeaturetrack_amd.compat import convolve as compat_convolve         # `from convolve import *` of a reference script
    assert compat_convolve._convolveSeparate is convolve._convolveSeparate
    f0 = img0.astype(np.float32)
    rows, cols = golden["img0_rows"], golden["img0_cols"]
    for name in (str(n) for n in golden["names"]):
        hk, vk = golden[name + "_h"], golden[name + "_v"]
        for tag in ("small", "tiny"):
            _same(convolve._convolveSeparate(golden[tag], list(hk), list(vk)), golden["%s_%s" % (name, tag)], "%s on %s" % (name, tag))
        out = convolve._convolveSeparate(f0, hk, vk)                         # (numpy arrays of taps work as lists do)
        _same(out[rows], golden[name + "_img0_rows"], name + " rows of img0")
        _same(np.ascontiguousarray(out[:, cols]), golden[name + "_img0_cols"], name + " columns of img0")
        assert hashlib.sha256(out.tobytes()).digest() == golden[name + "_img0_sha256"].tobytes(), name + ": sha256 of the whole output"


def test_smoothed_image_and_gradients_are_convolve_separate_calls(golden, cfg1, img0):
    """convolve.py:226-264: KLTComputeSmoothedImage = _convolveSeparate(img, gauss, gauss); gradx = (deriv, gauss), grady = (gauss, deriv)"""
    from pyfeaturetrack_amd import convolve
    g07, _ = convolve._computeKernels(0.1 * 7)
    f0 = img0.astype(np.float32)
    sm = convolve._convolveSeparate(f0, g07, g07)
    _same(sm, cfg1["sel_smooth"], "smooth(img0)")
    _same(sm, convolve.KLTComputeSmoothedImage(f0, 0.7), "KLTComputeSmoothedImage")
    g, d = convolve._computeKernels(1.0)
    _same(convolve._convolveSeparate(sm, d, g), cfg1["sel_gx"], "gradx(img0)")
    _same(convolve._convolveSeparate(sm, g, d), cfg1["sel_gy"], "grady(img0)")


@pytest.mark.parametrize("seed", range(6))
def test_convolve_separate_random_shapes_vs_oracle(seed):
    from oracle import klt_oracle as ko
    from pyfeaturetrack_amd import convolve
    rng = np.random.default_rng(900 + seed)
    for _ in range(12):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        if seed == 5:
            h, w = int(rng.integers(200, 700)), int(rng.integers(200, 900))
        img = (rng.random((h, w)) * 255).astype(np.float32)
        nh, nv = int(rng.integers(1, 72)), int(rng.integers(1, 72))
        hk, vk = rng.normal(size=nh), rng.normal(size=nv)
        kind = int(rng.integers(0, 4))
        if kind == 1:
            hk = (hk + hk[::-1]) / 2
        elif kind == 2:
            vk = (vk - vk[::-1]) / 2
        _same(convolve._convolveSeparate(img, hk, vk), ko.convolve_separate(img, hk, vk), "%dx%d image, %d / %d taps, kind %d" % (w, h, nh, nv, kind))


def test_convolve_separate_refuses_what_it_cannot_do():
    from pyfeaturetrack_amd import convolve
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    img = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError):
        convolve._convolveSeparate(img, [1.0] * 72, [1.0])
    with pytest.raises(ValueError):
        convolve._convolveSeparate(img, [], [1.0])
    with pytest.raises(ValueError):
        convolve._convolveSeparate(np.zeros((2, 3, 4), np.float32), [1.0], [1.0])
    ctx = Context(0)
    try:
        with pytest.raises(KltBackendError):
            ctx.convolve_separate(img, [1.0] * 72, [1.0])                     # the ABI checks for itself
        out = ctx.convolve_separate(img + 3, [1.0], [1.0])                    # ... and the context is usable afterwards
        assert np.array_equal(out, img + 3)
    finally:
        ctx.close()
