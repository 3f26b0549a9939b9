#!/usr/bin/env python3
"""Would ONE 4 MB copy per frame pair beat two 2 MB copies?  Copies only (klt_upload_u8_async into slots nothing
reads), `per` copies in flight before the host waits, next to a second context that builds and tracks resident pairs all the time --
the link as the pipelined ingest loop sees it.  Prints one JSON line."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import synth                                  # noqa: E402
from pyfeaturetrack_amd.backend import Context                        # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                # noqa: E402


def main():
    w, h = 1920, 1080
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    busy = Context(0)
    busy.configure(tc)
    f0, f1 = synth.synth_pair(w, h, 1)
    busy.upload(0, f0)
    busy.upload(1, f1)
    busy.build_pyramids_batch([0, 1], sync=True)
    fl, _ = busy.select(0, 5000, use_pyramid=True)
    busy.featbuf_upload(1, fl)
    stop = threading.Event()

    def compute():
        while not stop.is_set():
            for _ in range(8):
                busy.build_pyramids_batch([0, 1])
                busy.track_async(0, 1, 1, 2, 5000)
            busy.sync()

    c = Context(0)
    c.configure(tc)
    single = [c.pinned_array((h, w)) for _ in range(4)]
    double = [c.pinned_array((2 * h, w)) for _ in range(2)]
    for a in single + double:
        a[...] = 7
    out = {}
    for with_compute in (False, True):
        th = None
        if with_compute:
            stop.clear()
            th = threading.Thread(target=compute)
            th.start()
            time.sleep(0.05)
        for name in ("two copies of 2 MB per pair", "one copy of 4 MB per pair"):
            best = 0.0
            for rep in range(3):
                npairs = 400
                c.sync()
                t = time.perf_counter()
                for k in range(npairs):
                    if name.startswith("two"):
                        c.upload_async(10 + (2 * k) % 8, single[(2 * k) % 4])
                        c.upload_async(10 + (2 * k + 1) % 8, single[(2 * k + 1) % 4])
                    else:
                        c.upload_async(30 + k % 4, double[k % 2])
                    if k % 8 == 7:
                        c.upload_wait()
                c.upload_wait()
                dt = time.perf_counter() - t
                best = max(best, npairs * 2 * w * h / dt / 1e9)
            out[("next to build + track, " if with_compute else "idle GPU, ") + name] = round(best, 2)
        if th:
            stop.set()
            th.join()
    print(json.dumps(out))
    c.close()
    busy.close()


if __name__ == "__main__":
    main()
