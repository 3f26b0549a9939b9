"""Small helpers that live in the reference's klt_util.py.

KLTComputeSmoothSigma: klt_util.py:4-5.  KLTWriteFloatImageToPGM (klt_util.py:7-35) is a debug
dump that is broken in the reference (expects a PIL image, receives an ndarray); a working
ndarray version is provided under the same name.
"""
import numpy as np


def KLTComputeSmoothSigma(tc):
    return tc.smooth_sigma_fact * max(tc.window_width, tc.window_height)


def KLTWriteFloatImageToPGM(img, filename):
    from PIL import Image
    a = np.asarray(img, np.float32)
    lo, hi = float(a.min()), float(a.max())
    fact = 255.0 / (hi - lo) if hi != lo else 1.0
    Image.fromarray(((a - lo) * fact).astype(np.uint8), "L").save(filename)
