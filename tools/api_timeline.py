#!/usr/bin/env python3
"""Host-side timeline of one KLTTrackFeatures call at cfg-2 size: when each step of the call starts (us after the call began) and how
long the host spends in it, median over 100 calls, for the scenarios of tools/api_profile.py."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import _frames, backend, synth                 # noqa: E402
from pyfeaturetrack_amd import selectGoodFeatures as sgf               # noqa: E402
from pyfeaturetrack_amd import trackFeatures as tf                     # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                 # noqa: E402

W, H, N = 1920, 1080, 5000
LOG = []
T0 = [0.0]


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            LOG.append((label or name, (t - T0[0]) * 1e6, (time.perf_counter() - t) * 1e6))
    setattr(obj, name, timed)


def main():
    sgf.KLT_verbose = tf.KLT_verbose = 0
    for n in ("upload_async", "build_pyramids_batch", "build_pyramids", "track_enqueue", "track_complete", "configure", "swap_slots",
              "select_enqueue", "select_complete"):
        wrap(backend.Context, n)
    wrap(_frames, "same_pixels")
    wrap(_frames, "copy_pixels")
    wrap(tf, "features_to_array")
    wrap(tf, "shared_store")
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    base = synth.synth_base(W, H, 1)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 3.3, -2.1)
    g1 = f1.copy()
    clip = [synth.synth_frame(W, H, 1, k, base=base) for k in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))
    fl = sgf.KLTSelectGoodFeatures(tc, f0, N)

    def pingpong(k):
        a, b = (f0, f1) if k % 2 else (f1, f0)
        tf.KLTTrackFeatures(tc, a, b, fl)

    def fresh(k):
        g1[k % H, k % W] ^= 1
        tf.KLTTrackFeatures(tc, f0, g1, fl)

    def video(k):
        tf.KLTTrackFeatures(tc, clip[order[k % 30]], clip[order[(k + 1) % 30]], fl)

    tcs = KLT_TrackingContext()
    tcs.nPyramidLevels, tcs.subsampling = 3, 4
    tcs.KLTUpdateTCBorder()
    tcs.sequentialMode = True
    tcs.max_residue = 10.0
    fls = sgf.KLTSelectGoodFeatures(tcs, clip[0], N)
    wrap(sgf, "KLTCountRemainingFeatures")

    def sequential(k):
        cur = clip[order[(k + 1) % 30]]
        tf.KLTTrackFeatures(tcs, clip[order[k % 30]], cur, fls)
        sgf.KLTReplaceLostFeatures(tcs, cur, fls)

    for name, fn in (("ping-pong", pingpong), ("new frame 2 each call (one pixel)", fresh), ("consecutive frames of a clip", video),
                     ("sequential mode: KLTTrackFeatures + KLTReplaceLostFeatures", sequential)):
        for k in range(5):
            fn(k)
        runs, totals = [], []
        for k in range(5, 105):
            del LOG[:]
            T0[0] = time.perf_counter()
            fn(k)
            totals.append((time.perf_counter() - T0[0]) * 1e6)
            runs.append(list(LOG))
        shape = statistics.mode([tuple(r[0] for r in run) for run in runs])
        same = [run for run in runs if tuple(r[0] for r in run) == shape]
        print("== %s: %.1f us per call (median of 100; %d calls with the common step sequence)" % (name, statistics.median(totals), len(same)))
        for i, step in enumerate(shape):
            print("   %-22s starts %7.1f us   host time %6.1f us" % (step, statistics.median([r[i][1] for r in same]),
                                                                     statistics.median([r[i][2] for r in same])))


if __name__ == "__main__":
    main()
