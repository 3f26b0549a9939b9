"""How the colour test images are made from tests/golden/img0.pgm / img1.pgm -- the same three lines as in tests/golden/gen_colour_images.py
(kept here so that the tests do not import the generator, which needs the reference)."""
import numpy as np


def colour_of(grey):
    """uint8 [h, w] -> uint8 [h, w, 3]: three different functions of the frame, so that the luma is not the frame itself"""
    g = np.asarray(grey, np.uint8)
    return np.dstack([g, np.roll(g, 3, axis=1), (255 - g // 2).astype(np.uint8)])
