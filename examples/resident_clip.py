#!/usr/bin/env python3
"""A clip that is already in device memory (a hardware decoder's output, a cached recording): the frames are read IN PLACE
(klt_slot_adopt_u8 -- no upload, no copy), a ring of three frame slots, per frame the tracker, the replacement of lost features and --
on the context's build stream -- the pyramids and selection scores of the next frame.  The loop `bench.py --config cfg5` times, at a
size that runs in a second; it goes through the thin Context wrapper of the C ABI (include/klt_gpu.h), not the reference-shaped API.

    python examples/resident_clip.py [--frames 60] [--size 1280x720] [--features 2000]
"""
from __future__ import print_function

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                                    # noqa: E402

from pyfeaturetrack_amd import synth                                                  # noqa: E402
from pyfeaturetrack_amd.backend import Context, REPLACING_SOME, SELECTING_ALL         # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--features", type=int, default=2000)
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    n, npx = args.features, w * h

    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)

    clip = ctx.device_alloc(args.frames * npx)                       # stands in for frames a decoder left in device memory
    for k, f in enumerate(synth.periodic_sequence(w, h, 3, args.frames)):
        ctx.device_write(clip + k * npx, f)

    ring, lists = (0, 1, 2), (0, 1)
    ctx.set_option(15, 1)                                            # KLT_OPT_BUILD_STREAM: builds overlap the tracker / replacement

    def stage(k):                                                    # frame k: adopted in place, pyramids + selection scores enqueued
        ctx.adopt_u8(ring[k % 3], clip + k * npx, w, h)
        ctx.build_pyramids(ring[k % 3], sync=False)
        ctx.select_prepare(ring[k % 3])

    def track(k):                                                    # frame k-1 -> k
        ctx.track_async(ring[(k - 1) % 3], ring[k % 3], lists[(k - 1) % 2], lists[k % 2], n)

    t0 = time.perf_counter()
    stage(0)
    ctx.select_async(ring[0], SELECTING_ALL, True, lists[0], n)
    stage(1)
    track(1)
    for k in range(1, args.frames):
        ctx.select_begin(ring[k % 3], REPLACING_SOME, True, lists[k % 2], n)     # KLTReplaceLostFeatures, up to the host's look
        if k + 1 < args.frames:
            stage(k + 1)
            track(k + 1)                                             # only READS the list the selection completes
        if ctx.select_finish() and k + 1 < args.frames:
            track(k + 1)                                             # (rare) the selection rewrote the list after the tracker had read it
    last = ctx.featbuf_download(lists[(args.frames - 1) % 2], n)
    dt = time.perf_counter() - t0
    print("%d frames of %dx%d read in place from device memory, %d features: %.3f ms per frame; %d of %d alive at the end"
          % (args.frames, w, h, n, dt / (args.frames - 1) * 1e3, int((last["val"] >= 0).sum()), n))
    ctx.device_free(clip)
    ctx.close()


if __name__ == "__main__":
    main()
