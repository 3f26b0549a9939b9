#!/usr/bin/env python3
"""Independent frame pairs with several pairs in flight on one GPU (the configuration `bench.py` reports).

Frame pairs that do not depend on each other (BASELINE cfg-4: a batch of pairs; or the pairs of several cameras) can be
enqueued on different contexts of the same device.  Each context owns one HIP stream, its slots and its feature buffers;
nothing orders the streams against each other, so the GPU overlaps their kernels -- the drain / launch / ramp between the
dependent kernels of one pair, and the tracker's latency-bound wavefronts, are filled with the other pair's convolutions.

Frames arrive in pinned host memory (as a decoder would leave them); uploads run on each context's copy stream; the tracked
records of every pair land in a device-side table that is read back once per `--table` pairs.

    python examples/pairs_in_flight.py [--pairs 64] [--contexts 2] [--size 1920x1080] [--features 5000] [--frames-in-pinned-memory]

Measured on one MI355X (1080p, 5000 features): when the frames cross PCIe for every pair the link is the bound (two 2 MB frames =
0.09 ms) and one context is best -- 0.113 ms per pair with the frames already in pinned memory, 0.25 ms when the host also copies them
there; more contexts only make the copy streams compete (0.140 / 0.155 ms with 2 / 3).  More pairs in flight pay off when the
frames are already resident in HBM (`bench.py`: 0.058 -> 0.0415 ms per pair with three) -- e.g. produced on the device by a decoder or a
previous stage.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                # noqa: E402

from pyfeaturetrack_amd import synth                              # noqa: E402
from pyfeaturetrack_amd.backend import Context                    # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext            # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc              # noqa: E402

FB_SEL, FB_TABLE, FB_VIEW0 = 0, 1, 10


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--features", type=int, default=5000)
    ap.add_argument("--table", type=int, default=16, help="pairs per device-side record table (one download per table)")
    ap.add_argument("--frames-in-pinned-memory", action="store_true",
                    help="the frames already sit in the pinned buffers (skip the host-side copy that stands in for a decoder)")
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    n = args.features

    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    p = params_from_tc(tc)
    f0, f1 = synth.synth_pair(w, h, seed=1)

    ctxs = []
    for _ in range(args.contexts):
        cx = Context(0)
        cx.set_params(p)
        cx.upload(0, f0)
        cx.build_pyramids(0)
        fl, placed = cx.select(0, n, use_pyramid=True)            # features of frame 0 (every pair starts from them here)
        cx.featbuf_upload(FB_SEL, fl)
        cx.featbuf_alloc(FB_TABLE, args.table * n)
        for k in range(args.table):
            cx.featbuf_view(FB_VIEW0 + k, FB_TABLE, k * n, n)
        # two slot pairs and their pinned staging buffers, used alternately: the upload of a pair overlaps the kernels of
        # the previous pair of the same context
        cx.pins = {s: cx.pinned_array((h, w)) for s in (0, 1, 2, 3)}
        cx.done = 0
        ctxs.append(cx)

    results = []

    def submit(i):
        cx = ctxs[i % len(ctxs)]
        j = cx.done
        a = 0 if j % 2 == 0 else 2
        if not args.frames_in_pinned_memory or j < 2:
            cx.pins[a][:] = f0                                    # "decoder output"
            cx.pins[a + 1][:] = f1
        cx.upload_async(a, cx.pins[a])
        cx.upload_async(a + 1, cx.pins[a + 1])
        cx.build_pyramids_batch([a, a + 1])
        cx.track_async(a, a + 1, FB_SEL, FB_VIEW0 + j % args.table, n)
        cx.done += 1
        if cx.done % args.table == 0:
            results.append(cx.featbuf_download(FB_TABLE, args.table * n).reshape(args.table, n))

    for i in range(2 * len(ctxs)):                                # warm-up (allocations on first use)
        submit(i)
    for cx in ctxs:
        cx.sync()
        cx.done = 0
    results.clear()
    t = time.perf_counter()
    for i in range(args.pairs):
        submit(i)
    for cx in ctxs:
        cx.sync()
    dt = time.perf_counter() - t
    tracked = int((results[-1][-1]["val"] >= 0).sum()) if results else -1
    print("%d pairs %dx%d, %d features each, %d context(s): %.3f ms per pair (%sH2D of both frames and the record download included), "
          "%d tracked in the last pair" % (args.pairs, w, h, n, len(ctxs), dt / args.pairs * 1e3,
                                           "" if args.frames_in_pinned_memory else "host copies into pinned memory, ", tracked))
    for cx in ctxs:
        cx.close()


if __name__ == "__main__":
    main()
