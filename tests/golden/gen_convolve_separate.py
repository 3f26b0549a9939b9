#!/usr/bin/env python3
"""Golden vectors for `_convolveSeparate(imgin, horiz_kernel, vert_kernel)` (convolve.py:208-219), produced by RUNNING the reference
(development container only; reuses gen_golden.py's build step): the reference's one general separable convolution with the tap lists
the reference itself makes (gauss / gauss, deriv / gauss, gauss / deriv at three sigmas) and with tap lists it never makes -- neither
symmetric nor antisymmetric (the `sym == 0` branch of correlate1d), even counts (convolve1d then shifts the origin), a single tap, more
taps than the image has pixels.  Writes tests/golden/convolve_separate.npz: inputs and expected outputs, data only.

    python tests/golden/gen_convolve_separate.py
"""
import hashlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import build_reference  # noqa: E402

ROWS = (0, 1, 2, 119, 237, 238, 239)       # of img0's 240 rows: kept in full (the whole output by its sha256)
COLS = (0, 1, 2, 160, 317, 318, 319)


def main():
    refdir = build_reference()
    sys.path.insert(0, refdir)
    os.chdir(refdir)
    warnings.simplefilter("ignore")
    from PIL import Image
    import convolve as cv

    out = {}
    img0 = np.array(Image.open(os.path.join(HERE, "img0.pgm")).convert("F"))
    rng = np.random.default_rng(77)
    small = (rng.random((47, 61)) * 255).astype(np.float32)
    tiny = (rng.random((7, 9)) * 255).astype(np.float32)
    out["small"], out["tiny"] = small, tiny

    cases = []
    for sigma in (1.0, 0.7, 3.6):
        g, d = cv._computeKernels(sigma)
        g, d = np.array(g[:], np.float64), np.array(d[:], np.float64)
        cases += [("gauss_gauss_%g" % sigma, g, g), ("deriv_gauss_%g" % sigma, d, g), ("gauss_deriv_%g" % sigma, g, d)]
    r = np.random.default_rng(5)
    cases += [("asym_5_7", r.normal(size=5), r.normal(size=7)),                       # neither symmetric nor antisymmetric
              ("asym_9_sym_3", r.normal(size=9), np.array([0.25, 0.5, 0.25])),
              ("even_4_6", r.normal(size=4), r.normal(size=6)),                       # even counts: convolve1d shifts the origin
              ("even_2_odd_1", np.array([1.0, -1.0]), np.array([2.0])),               # forward difference, one tap
              ("one_one", np.array([1.0]), np.array([1.0])),                          # identity
              ("wide_13_11", r.normal(size=13), r.normal(size=11)),                   # wider than `tiny`: the reflection wraps more than once
              ("almost_sym", np.array([0.2, 0.5, 0.2 + 1e-15]), np.array([0.3, 0.4, 0.3 + 1e-12]))]   # inside / outside correlate1d's symmetry tolerance
    names = []
    for name, hk, vk in cases:
        names.append(name)
        out["%s_h" % name], out["%s_v" % name] = np.asarray(hk, np.float64), np.asarray(vk, np.float64)
        for tag, img in (("small", small), ("tiny", tiny)):
            res = cv._convolveSeparate(img, list(hk), list(vk))
            assert res.dtype == np.float32 and res.shape == img.shape
            out["%s_%s" % (name, tag)] = res
        res = cv._convolveSeparate(img0, list(hk), list(vk))
        assert res.dtype == np.float32
        out["%s_img0_sha256" % name] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(res).tobytes()).digest(), np.uint8)
        out["%s_img0_rows" % name] = res[list(ROWS)]
        out["%s_img0_cols" % name] = res[:, list(COLS)]
    out["names"] = np.array(names)
    out["img0_rows"], out["img0_cols"] = np.array(ROWS), np.array(COLS)
    np.savez_compressed(os.path.join(HERE, "convolve_separate.npz"), **out)
    print("wrote convolve_separate.npz: %d cases" % len(names))


if __name__ == "__main__":
    main()
