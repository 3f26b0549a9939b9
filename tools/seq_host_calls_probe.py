#!/usr/bin/env python3
"""Where does the HOST spend a frame of the 4K replace-every-frame sequence fed from pinned memory?  The loop of
tools/seq_contention_probe.py's `host` arrangement with a clock around every ABI call: per call the median / 90th percentile / maximum
over the frames and the mean over even and odd frames (the per-queue timeline shows a gap in every SECOND frame), for the upload issued
  after   select_finish (the product's order: the copy gets its head start while the re-run tracker is on the device), and
  early   right behind select_begin (a ring of four slots: an upload un-validates its slot's pyramids, and before the look the slot of
          frame k is still wanted by a repeated tracker); `after, ring of four` separates the ring's size from the order.
`--periodic`: bench.py's frames; `--table`: bench.py's read-back of 16 rows every 16 frames.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                     # noqa: E402

from pyfeaturetrack_amd import synth                                  # noqa: E402
from pyfeaturetrack_amd.backend import Context                        # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                # noqa: E402


def run(order_of_send, w, h, n, nframes):
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)
    clock = {}

    def timed(name, f, *a):
        t = time.perf_counter()
        r = f(*a)
        clock.setdefault(name, []).append((time.perf_counter() - t) * 1e6)
        return r

    try:
        NPIN = 16
        pins = []
        if "--periodic" in sys.argv:                                       # bench.py's frames (replacement passes are heavier on them)
            source = synth.periodic_sequence(w, h, 4, NPIN, phases=synth.sequence_phases(w, h, 4, workers=6))
        else:
            base = synth.synth_base(w, h, 4)
            source = (synth.synth_frame(w, h, 4, k, base=base) for k in range(NPIN))
        for f in source:
            a = ctx.pinned_array((h, w))
            a[:] = f
            pins.append(a)
        order = list(range(NPIN)) + list(range(NPIN - 2, 0, -1))
        # an upload un-validates its slot's pyramids: sent BEFORE the look it must not go into the slot of frame k, which a repeated tracker
        # (k -> k + 1) still reads -- one more slot in the ring then
        ring, ahead = (3 if order_of_send == "after" else 4), 2
        NT, TAB = 16, 100
        ctx.featbuf_alloc(TAB, 2 * NT * n)
        for k in range(2 * NT):
            ctx.featbuf_view(TAB + 1 + k, TAB, k * n, n)
        row = lambda k: TAB + 1 + k % (2 * NT)                         # noqa: E731
        table = "--table" in sys.argv                                      # bench.py's read-back of 16 rows every 16 frames
        if table:
            from pyfeaturetrack_amd.backend import FEAT_DTYPE
            HALF = (200, 201)
            for i in range(2):
                ctx.featbuf_view(HALF[i], TAB, i * NT * n, NT * n)
            host_tab = [ctx.pinned_array((NT * n,), FEAT_DTYPE) for _ in range(2)]
        ctx.set_option(15, 1)                                          # KLT_OPT_BUILD_STREAM

        def send(k):
            timed("upload_async", ctx.upload_async, k % ring, pins[order[k % len(order)]])

        def stage(k):
            timed("build_pyramids", lambda: ctx.build_pyramids(k % ring, sync=False))
            timed("select_prepare", ctx.select_prepare, k % ring)

        def track(k, name="track_async"):
            timed(name, ctx.track_async, (k - 1) % ring, k % ring, row(k - 1), row(k), n)

        def loop(count):
            send(0)
            ctx.build_pyramids(0, sync=False)
            ctx.select_async(0, 1, True, row(0), n)
            for j in range(1, ahead + 1):
                send(j)
            stage(1)
            track(1)
            send(ahead + 1)
            for k in range(1, count):
                timed("select_begin", ctx.select_begin, k % ring, 2, True, row(k), n)
                if order_of_send == "early":
                    send(k + ahead + 1)
                stage(k + 1)
                track(k + 1)
                if timed("select_finish", ctx.select_finish):
                    track(k + 1, "track_async (again)")
                if order_of_send != "early":
                    send(k + ahead + 1)
                if table and k % NT == NT - 1:
                    ctx.download_wait()
                    ctx.featbuf_download_async(HALF[(k // NT) % 2], host_tab[(k // NT) % 2])
            if table:
                ctx.download_wait()
            ctx.sync()

        loop(32)
        best = None
        for _ in range(3):
            clock.clear()
            t = time.perf_counter()
            loop(nframes)
            ms = (time.perf_counter() - t) / (nframes - 1) * 1e3
            best = ms if best is None else min(best, ms)
        calls = {}
        for name, v in clock.items():
            v = np.array(v[4:] if len(v) > 8 else v)
            calls[name] = {"median": round(float(np.median(v)), 1), "p90": round(float(np.percentile(v, 90)), 1), "max": round(float(v.max()), 1),
                           "mean_even": round(float(v[0::2].mean()), 1), "mean_odd": round(float(v[1::2].mean() if v.size > 1 else v.mean()), 1), "calls": int(v.size)}
        return {"ms_per_frame": round(best, 4), "host_us_per_call": calls,
                "host_us_per_frame_outside_select_finish": round(sum(np.sum(v) for k, v in clock.items() if k != "select_finish") / (nframes - 1), 1)}
    finally:
        ctx.close()


def main():
    w, h, n, nframes = (3840, 2160, 20000, 128) if "--1080p" not in sys.argv else (1920, 1080, 5000, 256)
    out = {"frame": "%dx%d" % (w, h), "features": n, "frames": nframes, "flags": [a for a in sys.argv[1:]]}
    for rep in range(int(os.environ.get("KLT_PROBE_REPS", "4"))):
        for mode in ("after", "after, ring of four", "early"):
            out.setdefault("upload " + mode, []).append(run(mode, w, h, n, nframes))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
