#!/usr/bin/env python3
"""Why does a 4K replace-every-frame sequence fed from host memory run slower than the same sequence on resident frames (0.28-0.30 against
0.23 ms per frame)?  The ABI loop of bench.py's `sequence_from_host` in four arrangements on one box:

  resident        frames already in device memory, adopted in place (klt_slot_adopt_u8): no copy at all
  resident+copies the same, PLUS one 8 MB host-to-device copy per frame into a slot nothing ever reads: the link and the copy engines are
                  as busy as in `host`, but no kernel depends on a copy
  host            one new frame per step from pinned memory, sent two steps ahead (the product's arrangement)
  host, 3 ahead   sent three steps ahead (a ring of four slots)

If `resident+copies` reads like `resident`, the loss of `host` is the dependency chain (copy -> build); if it reads like `host`, it is
what a running 53 GB/s copy costs the kernels next to it.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import synth                                  # noqa: E402
from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE            # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                # noqa: E402


def run(mode, w, h, n, nframes):
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)
    try:
        NPIN = 16
        base = synth.synth_base(w, h, 4)
        pins, devs = [], []
        for k in range(NPIN):
            f = synth.synth_frame(w, h, 4, k, base=base)
            a = ctx.pinned_array((h, w))
            a[:] = f
            pins.append(a)
            if mode.startswith("resident"):
                d = ctx.device_alloc(f.nbytes)
                ctx.device_write(d, f)
                devs.append(d)
        order = list(range(NPIN)) + list(range(NPIN - 2, 0, -1))
        ring = 4 if mode == "host, 3 ahead" else 3
        ahead = ring - 1
        NT = 16
        TAB = 100
        ctx.featbuf_alloc(TAB, 2 * NT * n)
        for k in range(2 * NT):
            ctx.featbuf_view(TAB + 1 + k, TAB, k * n, n)
        row = lambda k: TAB + 1 + k % (2 * NT)                         # noqa: E731
        ctx.set_option(15, 1)                                          # KLT_OPT_BUILD_STREAM

        def send(k):
            i = order[k % len(order)]
            if mode.startswith("resident"):
                ctx.adopt_u8(k % ring, devs[i], w, h)
                if mode == "resident+copies":
                    ctx.upload_async(50 + k % 2, pins[i])
            else:
                ctx.upload_async(k % ring, pins[i])

        def stage(k):
            ctx.build_pyramids(k % ring, sync=False)
            ctx.select_prepare(k % ring)

        def track(k):
            ctx.track_async((k - 1) % ring, k % ring, row(k - 1), row(k), n)

        def loop(count):
            if mode.startswith("resident"):
                # an adopted frame is there at once: the slot is (re)pointed right before its build
                send(0)
                ctx.build_pyramids(0, sync=False)
                ctx.select_async(0, 1, True, row(0), n)
                send(1)
                stage(1)
                track(1)
                for k in range(1, count):
                    ctx.select_begin(k % ring, 2, True, row(k), n)
                    send(k + 1)
                    stage(k + 1)
                    track(k + 1)
                    if ctx.select_finish():
                        track(k + 1)
            else:
                send(0)
                ctx.build_pyramids(0, sync=False)
                ctx.select_async(0, 1, True, row(0), n)
                for j in range(1, ahead + 1):
                    send(j)
                stage(1)
                track(1)
                send(ahead + 1)
                for k in range(1, count):
                    ctx.select_begin(k % ring, 2, True, row(k), n)
                    stage(k + 1)
                    track(k + 1)
                    if ctx.select_finish():
                        track(k + 1)
                    send(k + ahead + 1)
            ctx.sync()

        loop(2 * NT)
        best = None
        for _ in range(3):
            t = time.perf_counter()
            loop(nframes)
            ms = (time.perf_counter() - t) / (nframes - 1) * 1e3
            best = ms if best is None else min(best, ms)
        alive = int((ctx.featbuf_download(row(nframes - 1), n)["val"] >= 0).sum())
        return {"ms_per_frame": best, "alive": alive}
    finally:
        ctx.close()


def main():
    w, h, n, nframes = (3840, 2160, 20000, 128) if "--1080p" not in sys.argv else (1920, 1080, 5000, 256)
    out = {"frame": "%dx%d" % (w, h), "features": n, "frames": nframes}
    for rep in range(2):
        for mode in (os.environ.get("KLT_PROBE_MODES", "resident,resident+copies,host,host, 3 ahead").split(",") if os.environ.get("KLT_PROBE_MODES") else ("resident", "resident+copies", "host", "host, 3 ahead")):
            r = run(mode, w, h, n, nframes)
            out.setdefault(mode, []).append(round(r["ms_per_frame"], 4))
            out["alive_" + mode] = r["alive"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
