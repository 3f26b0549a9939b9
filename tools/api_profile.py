#!/usr/bin/env python3
"""Where the host time of a reference-shaped API call goes (cfg-2 size: 1920x1080, 5000 features): cProfile tables of
KLTSelectGoodFeatures, KLTTrackFeatures (ping-pong on two resident frames; a new second frame in every call) and the plain
wall-clock medians next to them, plus what the host primitives under them cost on this box (memcmp / copy of one frame,
creation of 5000 feature objects).  Writes text to stdout; `tools/api_probe.py` prints the one-line JSON summary."""
import cProfile
import ctypes
import io
import os
import pstats
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import selectGoodFeatures as sgf          # noqa: E402
from pyfeaturetrack_amd import synth                               # noqa: E402
from pyfeaturetrack_amd import trackFeatures as tf                  # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext, new_feature_list  # noqa: E402

W, H, N = 1920, 1080, 5000


def med(fn, reps=40):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts) * 1e3


def table(fn, reps, title, rows=18):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(rows)
    body = s.getvalue()
    print("=== %s: cProfile over %d calls (tottime; the profiler's own overhead inflates Python-level rows)" % (title, reps))
    print(body[body.index("ncalls") - 3:])


def main():
    sgf.KLT_verbose = tf.KLT_verbose = 0
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    base = synth.synth_base(W, H, 1)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 3.3, -2.1)
    g1 = f1.copy()
    state = {"fl": sgf.KLTSelectGoodFeatures(tc, f0, N), "k": 0}
    tf.KLTTrackFeatures(tc, f0, f1, state["fl"])

    def select():
        state["fl"] = sgf.KLTSelectGoodFeatures(tc, f0, N)

    def pingpong():
        k = state["k"] = state["k"] + 1
        a, b = (f0, f1) if k % 2 else (f1, f0)
        tf.KLTTrackFeatures(tc, a, b, state["fl"])

    def fresh():
        k = state["k"] = state["k"] + 1
        g1[k % H, k % W] ^= 1
        tf.KLTTrackFeatures(tc, f0, g1, state["fl"])

    clip = [synth.synth_frame(W, H, 1, k, base=base) for k in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))

    def video():
        # consecutive frames of a clip, non-sequential mode: frame 1 is the previous call's frame 2, frame 2 has new pixels
        k = state["k"] = state["k"] + 1
        tf.KLTTrackFeatures(tc, clip[order[k % 30]], clip[order[(k + 1) % 30]], state["fl"])

    print("host primitives on this box (ms):")
    a, b = f0.copy(), f0.copy()
    memcmp = ctypes.CDLL(None).memcmp
    memcmp.argtypes = (ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
    print("  memcmp of two equal %dx%d u8 frames        %.4f" % (W, H, med(lambda: memcmp(a.ctypes.data, b.ctypes.data, a.nbytes))))
    print("  numpy copy of one frame                      %.4f" % med(lambda: np.copyto(b, a)))
    print("  new_feature_list(%d)                        %.4f" % (N, med(lambda: new_feature_list(N))))
    print("wall clock per call (ms, median of 40):")
    for name, fn in (("KLTSelectGoodFeatures", select), ("KLTTrackFeatures ping-pong", pingpong), ("KLTTrackFeatures new frame 2 each call", fresh),
                     ("KLTTrackFeatures consecutive frames of a clip", video)):
        fn()
        print("  %-46s %.4f" % (name, med(fn)))
    for name, fn in (("KLTSelectGoodFeatures", select), ("KLTTrackFeatures ping-pong", pingpong), ("KLTTrackFeatures new frame 2 each call", fresh),
                     ("KLTTrackFeatures consecutive frames of a clip", video)):
        table(fn, 200, name)


if __name__ == "__main__":
    main()
