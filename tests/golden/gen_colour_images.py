#!/usr/bin/env python3
"""Golden vectors for colour and float Pillow images on the API path, produced by RUNNING the reference (development container only; reuses
gen_golden.py's build step): KLTSelectGoodFeatures / KLTTrackFeatures of the reference on "RGB", "RGBA" and "F" images -- the reference
converts whatever it is given with `img.convert("F")` (selectGoodFeatures.py:190,194, trackFeatures.py:165,176), i.e. Pillow's ITU-R 601-2
luma for colour images.  The images are made from tests/golden/img0.pgm / img1.pgm by `colour_of` below (the tests rebuild them with the
same three lines), so only the expected lists are stored.  Writes tests/golden/colour_images.npz.

    python tests/golden/gen_colour_images.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import build_reference  # noqa: E402


def colour_of(grey):
    """uint8 [h, w] -> uint8 [h, w, 3]: three different functions of the frame, so that the luma is not the frame itself"""
    g = np.asarray(grey, np.uint8)
    return np.dstack([g, np.roll(g, 3, axis=1), (255 - g // 2).astype(np.uint8)])


def main():
    refdir = build_reference()
    sys.path.insert(0, refdir)
    os.chdir(refdir)
    warnings.simplefilter("ignore")
    from PIL import Image
    import klt
    import selectGoodFeatures as sgf
    import trackFeatures as tf
    sgf.KLT_verbose = tf.KLT_verbose = 0
    g0 = np.array(Image.open(os.path.join(HERE, "img0.pgm")))
    g1 = np.array(Image.open(os.path.join(HERE, "img1.pgm")))
    c0, c1 = colour_of(g0), colour_of(g1)
    alpha = (np.arange(g0.size, dtype=np.uint32).reshape(g0.shape) * 7 % 256).astype(np.uint8)
    out = {}
    cases = {
        "rgb": (Image.fromarray(c0, "RGB"), Image.fromarray(c1, "RGB")),
        "rgba": (Image.fromarray(np.dstack([c0, alpha]), "RGBA"), Image.fromarray(np.dstack([c1, alpha]), "RGBA")),
        "f": (Image.fromarray(c0, "RGB").convert("F"), Image.fromarray(c1, "RGB").convert("F")),
    }
    for name, (i0, i1) in cases.items():
        tc = klt.KLT_TrackingContext()
        tc.max_residue = 10.0
        fl = sgf.KLTSelectGoodFeatures(tc, i0, 100)
        out[name + "_sel"] = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
        tf.KLTTrackFeatures(tc, i0, i1, fl)
        out[name + "_trk"] = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
        tf.KLTTrackFeatures(tc, i1, i0, fl)                                 # ... and back (example1's ping-pong)
        out[name + "_back"] = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
    assert np.array_equal(out["rgb_trk"], out["rgba_trk"]) and np.array_equal(out["rgb_trk"], out["f_trk"])
    out["luma0_rows"] = np.array(cases["rgb"][0].convert("F"))[[0, 119, 239]]   # three rows of Pillow's float frame, for the conversion itself
    np.savez_compressed(os.path.join(HERE, "colour_images.npz"), **out)
    print("wrote colour_images.npz: tracked %d of 100" % int((out["rgb_trk"][:, 2] >= 0).sum()))


if __name__ == "__main__":
    main()
