#!/usr/bin/env python3
"""Track a synthetic sequence with the reference-shaped API in sequential mode, replacing lost features after every
frame (BASELINE cfg-5's call pattern, at a size that runs in a second): KLTSelectGoodFeatures once, then per frame
KLTTrackFeatures + KLTReplaceLostFeatures.  The frame-2 pyramids of a call stay on the device and become frame 1 of the
next call; replacement reuses their level-0 image and gradients.

    python examples/sequence.py [--frames 30] [--size 1280x720] [--features 2000] [--affine]
"""
from __future__ import print_function

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                                    # noqa: E402

from pyfeaturetrack_amd import selectGoodFeatures as sgf                              # noqa: E402
from pyfeaturetrack_amd import synth                                                  # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext, KLTCountRemainingFeatures     # noqa: E402
from pyfeaturetrack_amd import trackFeatures as tf                                    # noqa: E402
from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--affine", action="store_true", help="switch the affine consistency check on (mode 2)")
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))

    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.sequentialMode = True
    tc.max_residue = 10.0
    if args.affine:
        tc.affineConsistencyCheck = 2
    sgf.KLT_verbose = tf.KLT_verbose = 0            # (each module binds its own switch at import, as the reference's do)

    base = synth.synth_base(w, h, seed=3)
    frame = lambda k: synth.synth_frame(w, h, 3, k, base=base)          # noqa: E731  (numpy uint8 frames are accepted)
    prev = frame(0)
    fl = sgf.KLTSelectGoodFeatures(tc, prev, args.features)
    start = np.array([(f.x, f.y) for f in fl], np.float64)
    t0 = time.perf_counter()
    replaced = 0
    for k in range(1, args.frames):
        cur = frame(k)
        KLTTrackFeatures(tc, prev, cur, fl)
        lost = len(fl) - KLTCountRemainingFeatures(fl)
        replaced += lost
        sgf.KLTReplaceLostFeatures(tc, cur, fl)
        prev = cur
    dt = time.perf_counter() - t0
    survivors = [i for i, f in enumerate(fl) if f.val == 0]
    moved = np.array([(fl[i].x, fl[i].y) for i in survivors]) - start[survivors]
    print("%d frames of %dx%d, %d features: %.2f ms per frame (host API, frame generation included); "
          "%d replacements; survivors moved by (%.2f, %.2f) px per frame (imposed %.1f, %.1f)"
          % (args.frames - 1, w, h, len(fl), dt / (args.frames - 1) * 1e3, replaced,
             np.median(moved[:, 0]) / (args.frames - 1), np.median(moved[:, 1]) / (args.frames - 1),
             synth.DEFAULT_SHIFT[0], synth.DEFAULT_SHIFT[1]))


if __name__ == "__main__":
    main()
