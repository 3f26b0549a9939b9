#!/usr/bin/env python3
"""Select the 50 best features of img0.pgm, track them back and forth between img0.pgm and img1.pgm,
print them and write feat1.ppm / feat2.ppm -- this repository's counterpart of the reference's
example1.py driver (same call sequence and print format), on the MI355X backend.

    python examples/example1.py [--iterations 100] [--dir tests/golden]
"""
from __future__ import print_function

import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from PIL import Image                                                     # noqa: E402

from pyfeaturetrack_amd.klt import KLT_TrackingContext, KLTPrintTrackingContext   # noqa: E402
from pyfeaturetrack_amd.selectGoodFeatures import KLTSelectGoodFeatures            # noqa: E402
from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures                      # noqa: E402
from pyfeaturetrack_amd.writeFeatures import KLTWriteFeatureListToPPM              # noqa: E402


def show(title, fl):
    print("\n" + title)
    for i, feat in enumerate(fl):
        print("Feature #{0}:  ({1},{2}) with value of {3}".format(i, feat.x, feat.y, feat.val))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--out", default=".")
    args = ap.parse_args()

    tc = KLT_TrackingContext()
    nFeatures = 50
    tc.nSkippedPixels = 0
    tc.max_residue = 10.0
    KLTPrintTrackingContext(tc)

    img1 = Image.open(os.path.join(args.dir, "img0.pgm"))
    img2 = Image.open(os.path.join(args.dir, "img1.pgm"))

    fl = KLTSelectGoodFeatures(tc, img1, nFeatures)
    show("In first image:", fl)
    KLTWriteFeatureListToPPM(fl, img1, os.path.join(args.out, "feat1.ppm"))

    calls = 0
    t0 = time.perf_counter()
    for _ in range(args.iterations):
        KLTTrackFeatures(tc, img1, img2, fl)
        KLTTrackFeatures(tc, img2, img1, fl)
        calls += 2
    print((time.perf_counter() - t0) / max(calls, 1))

    show("In second image:", fl)
    KLTWriteFeatureListToPPM(fl, img2, os.path.join(args.out, "feat2.ppm"))


if __name__ == "__main__":
    main()
