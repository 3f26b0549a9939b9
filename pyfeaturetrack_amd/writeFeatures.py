"""Feature-list output (reference: writeFeatures.py).  Host-side Python only."""
from __future__ import print_function

import numpy as np

from .klt import KLTCountRemainingFeatures
from .selectGoodFeatures import KLT_verbose       # bound at import, as writeFeatures.py:3 does: this module's own switch


def KLTWriteFeatureListToPPM(featurelist, greyimg, filename):
    """Overlay every live feature as a red 3x3 square on an RGB copy and save it
    (writeFeatures.py:10-37; the square is centred on int(x + 0.5), int(y + 0.5))."""
    ncols, nrows = greyimg.size
    if KLT_verbose:
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


_BINHEADER_FL = b"KLTFL1"
_VAL_WIDTH = 5


def _format_width(fmt):
    """Printed width of one `(x,y)=val ` group (upstream _findStringWidth: the digits of every conversion plus the literals)."""
    import re
    group = "(%s,%s)=%%%dd " % (fmt, fmt, _VAL_WIDTH)
    width = 0
    for lit, w in re.findall(r"([^%]+)|%(\d*)(?:\.\d+)?[a-zA-Z]", group):
        width += len(lit) if lit else int(w or 1)
    return width


def KLTWriteFeatureList(featurelist, filename, fmt="%5.1f"):
    """Feature list to a text file (fmt such as "%5.1f" or "%3d") or, with fmt None, to a binary file.

    The reference's function (writeFeatures.py:53-82) calls helpers it never defines (_printSetupTxt, _printHeader,
    FEATURE_LIST ...) and cannot run; the layout here is upstream KLT 1.3.4's: a comment area, the "do not modify" line, the
    `KLT Feature List` banner, `nFeatures = n`, one `%7d | (x,y)=val ` row per feature; an integer format prints the
    rounded position.  Binary: b"KLTFL1", int32 nFeatures, then (float32 x, float32 y, int32 val) per feature."""
    fmt_str = "binary" if fmt is None else "text"
    if KLT_verbose >= 1 and filename is not None:
        print("(KLT) Writing feature list to {0} file: '{1}'".format(fmt_str, filename))
    n = len(featurelist)
    if fmt is None:
        rec = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("val", "<i4")]))
        rec["x"] = [f.x for f in featurelist]
        rec["y"] = [f.y for f in featurelist]
        rec["val"] = [f.val for f in featurelist]
        with open(filename, "wb") as f:
            f.write(_BINHEADER_FL)
            f.write(np.int32(n).tobytes())
            f.write(rec.tobytes())
        return
    integer = fmt.rstrip()[-1] == "d"
    if not integer and fmt.rstrip()[-1] != "f":
        from .error import KLTError
        KLTError("(KLTWriteFeatureList) Bad format: {0}".format(fmt))
    group = "(%s,%s)=%%%dd " % (fmt, fmt, _VAL_WIDTH)
    with open(filename, "w") as f:
        f.write("Feel free to place comments here.\n\n\n")
        f.write("!!! Warning:  This is a KLT data file.  Do not modify below this line !!!\n")
        f.write("\n------------------------------\nKLT Feature List\n------------------------------\n\n")
        f.write("nFeatures = %d\n\n" % n)
        f.write("feature | (x,y)=val\n--------+-" + "-" * _format_width(fmt) + "\n")
        for i, feat in enumerate(featurelist):
            if integer:
                x, y = int(feat.x + 0.5), int(feat.y + 0.5)      # upstream rounds for integer formats
            else:
                x, y = feat.x, feat.y
            f.write("%7d | " % i + group % (x, y, feat.val) + "\n")


def KLTReadFeatureList(filename):
    """Reads what KLTWriteFeatureList wrote (either form) back into a list of KLT_Feature."""
    import re
    from .klt import KLT_Feature
    with open(filename, "rb") as f:
        data = f.read()
    out = []
    if data.startswith(_BINHEADER_FL):
        n = int(np.frombuffer(data, "<i4", 1, len(_BINHEADER_FL))[0])
        rec = np.frombuffer(data, np.dtype([("x", "<f4"), ("y", "<f4"), ("val", "<i4")]), n, len(_BINHEADER_FL) + 4)
        rows = zip(rec["x"].tolist(), rec["y"].tolist(), rec["val"].tolist())
    else:
        text = data.decode()
        if "KLT Feature List" not in text:
            from .error import KLTError
            KLTError("(KLTReadFeatureList) File '{0}' does not contain a feature list".format(filename))
        body = text[text.index("--------+-"):]
        rows = [(float(m.group(1)), float(m.group(2)), int(m.group(3)))
                for m in re.finditer(r"^\s*\d+ \| \(\s*([-\d.]+),\s*([-\d.]+)\)=\s*(-?\d+)", body, flags=re.M)]
    for x, y, v in rows:
        feat = KLT_Feature()
        feat.x, feat.y, feat.val = x, y, int(v)
        out.append(feat)
    return out


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
