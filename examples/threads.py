#!/usr/bin/env python3
"""Several videos tracked at the same time, one Python thread each, through the reference-shaped API: every thread gets its own device
context (its own HIP streams, parameters and pinned buffers; `backend.default_context`) and every KLT* call holds that context's lock,
so threads cannot see each other's frames, parameters or records.  Prints frames per second with the clips one after the other and with
`--threads` threads at once, and checks that every thread's final list equals what the same clip gives on its own.

What to expect: SAFE, not faster.  A per-frame loop at 1080p spends more than half of a call in the interpreter (column moves, frame
keys, ctypes), which the interpreter's lock serialises; only the waits for the device overlap, and four threads read 0.9x of the
one-after-the-other rate.  Throughput comes from `KLTTrackSequence` (one call per clip: 0.15 ms per 1080p frame) or from batched
launches (`examples/batched_pairs.py`), not from threads.

    python examples/threads.py [--threads 4] [--frames 120] [--size 1920x1080] [--features 5000]
"""
from __future__ import print_function

import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import selectGoodFeatures as sgf                              # noqa: E402
from pyfeaturetrack_amd import synth                                                  # noqa: E402
from pyfeaturetrack_amd import trackFeatures as tf                                    # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                                # noqa: E402


def make_clip(w, h, seed, distinct=12):
    base = synth.synth_base(w, h, seed)
    frames = [synth.synth_frame(w, h, seed, k, base=base) for k in range(distinct)]
    return frames + frames[-2:0:-1]                       # up and down: consecutive frames always differ by one step


def track_clip(clip, nframes, nfeatures):
    """the loop a script written against the reference runs: select once, then per frame track + replace the lost features"""
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.sequentialMode = True
    tc.max_residue = 10.0
    fl = sgf.KLTSelectGoodFeatures(tc, clip[0], nfeatures)
    for k in range(1, nframes):
        prev, cur = clip[(k - 1) % len(clip)], clip[k % len(clip)]
        tf.KLTTrackFeatures(tc, prev, cur, fl)
        sgf.KLTReplaceLostFeatures(tc, cur, fl)
    return [(f.x, f.y, f.val) for f in fl]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--features", type=int, default=5000)
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    sgf.KLT_verbose = tf.KLT_verbose = 0
    clips = [make_clip(w, h, 10 + i) for i in range(args.threads)]

    track_clip(clips[0], 8, args.features)                 # opens the device, loads the kernels
    t = time.perf_counter()
    alone = [track_clip(c, args.frames, args.features) for c in clips]
    t_serial = time.perf_counter() - t

    together, errors = [None] * args.threads, []

    def work(i):
        try:
            together[i] = track_clip(clips[i], args.frames, args.features)
        except BaseException as e:                          # noqa: BLE001 -- reported below
            errors.append(e)

    for i in range(args.threads):                           # each thread's first call opens its own context: not timed
        th = threading.Thread(target=lambda i=i: track_clip(clips[i], 4, args.features))
        th.start()
        th.join()
    workers = [threading.Thread(target=work, args=(i,)) for i in range(args.threads)]
    t = time.perf_counter()
    for th in workers:
        th.start()
    for th in workers:
        th.join()
    t_threads = time.perf_counter() - t
    if errors:
        raise errors[0]
    same = all(a == b for a, b in zip(alone, together))
    total = args.threads * (args.frames - 1)
    print("%d clips of %d frames, %dx%d, %d features each" % (args.threads, args.frames, w, h, args.features))
    print("one after the other : %.1f frames/s (%.3f ms per frame)" % (total / t_serial, t_serial / total * 1e3))
    print("%d threads at once   : %.1f frames/s (%.3f ms per frame), %.2fx" % (args.threads, total / t_threads, t_threads / total * 1e3,
                                                                              t_serial / t_threads))
    print("every thread's final list equals its clip's own: %s" % same)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
