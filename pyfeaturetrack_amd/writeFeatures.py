"""Feature-list output (reference: writeFeatures.py).  Host-side Python only."""
from __future__ import print_function

import numpy as np

from . import selectGoodFeatures as _sgf
from .klt import KLTCountRemainingFeatures


def KLTWriteFeatureListToPPM(featurelist, greyimg, filename):
    """Overlay every live feature as a red 3x3 square on an RGB copy and save it
    (writeFeatures.py:10-37; the square is centred on int(x + 0.5), int(y + 0.5))."""
    ncols, nrows = greyimg.size
    if _sgf.KLT_verbose:
        print("(KLT) Writing {0} features to PPM file: '{1}'".format(KLTCountRemainingFeatures(featurelist), filename))
    rgb = np.array(greyimg.convert("RGB"))
    for feat in featurelist:
        if feat.val >= 0:
            x = int(feat.x + 0.5)
            y = int(feat.y + 0.5)
            x0, x1 = max(x - 1, 0), min(x + 2, ncols)
            y0, y1 = max(y - 1, 0), min(y + 2, nrows)
            if x0 < x1 and y0 < y1:
                rgb[y0:y1, x0:x1] = (255, 0, 0)
    from PIL import Image
    Image.fromarray(rgb, "RGB").save(filename)


def KLTWriteFeatureList(featurelist, filename, fmt="%5.1f"):
    """Text dump of a feature list.  The reference's version (writeFeatures.py:53-82) references
    undefined helpers and cannot run; this writes one `index x y val` line per feature."""
    with open(filename, "w") as f:
        f.write("# KLT feature list: %d features\n" % len(featurelist))
        for i, feat in enumerate(featurelist):
            f.write("%d %s %s %d\n" % (i, fmt % feat.x, fmt % feat.y, feat.val))


def KLTWriteFeatureTable(ft, filename, fmt="%5.1f"):
    """Text dump of a KLT_FeatureTable (upstream KLTWriteFeatureTable's text form: one line per feature, one
    `(x,y)=val` group per frame)."""
    with open(filename, "w") as f:
        f.write("# KLT feature table: %d frames x %d features\n" % (ft.nFrames, ft.nFeatures))
        xs, ys, vs = ft.x.T.tolist(), ft.y.T.tolist(), ft.val.T.tolist()
        for i in range(ft.nFeatures):
            f.write("%d |" % i)
            for x, y, v in zip(xs[i], ys[i], vs[i]):
                f.write(" (%s,%s)=%d" % (fmt % x, fmt % y, v))
            f.write("\n")
