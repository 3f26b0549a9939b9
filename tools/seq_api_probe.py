#!/usr/bin/env python3
"""KLTTrackSequence at 4K (20 000 features, replacement after every frame) from numpy frames: ms per frame, and where the calling thread
waits -- for the helper thread's staged frame (`_FrameStager.next`), in klt_select_finish, in the other library calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import backend, synth, trackSequence                # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                       # noqa: E402

ACC = {}


def wrap(obj, name):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = ACC.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t
    setattr(obj, name, timed)


def main():
    w, h, n, nframes = (3840, 2160, 20000, 128) if "--1080p" not in sys.argv else (1920, 1080, 5000, 256)
    for name in ("select_finish", "select_begin", "build_pyramids", "select_prepare", "track_async", "upload_async", "featbuf_download_into",
                 "upload_wait", "select_async"):
        wrap(backend.Context, name)
    wrap(trackSequence._FrameStager, "next")
    if "--raw" in sys.argv:
        lib = backend.load_library()

        class Lib(object):
            def __getattr__(self, name):
                return getattr(lib, name)
        proxy = Lib()
        for name in ("klt_upload_u8_async", "klt_select_begin_async", "klt_build_pyramids_async"):
            fn = getattr(lib, name)

            def timed(*a, _fn=fn, _name=name):
                t = time.perf_counter()
                try:
                    return _fn(*a)
                finally:
                    e = ACC.setdefault("  lib." + _name, [0, 0.0])
                    e[0] += 1
                    e[1] += time.perf_counter() - t
            setattr(proxy, name, timed)
        orig_init = backend.Context.__init__

        def init(self, *a, **k):
            orig_init(self, *a, **k)
            self._lib = proxy
        backend.Context.__init__ = init
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    base = synth.synth_base(w, h, 4)
    distinct = [synth.synth_frame(w, h, 4, k, base=base) for k in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))
    frames = [distinct[order[k % 30]] for k in range(nframes)]
    for rep in range(3):
        ACC.clear()
        t = time.perf_counter()
        trackSequence.KLTTrackSequence(tc, frames, n)
        total = time.perf_counter() - t
        print("run %d: %.4f ms per frame" % (rep, total * 1e3 / (nframes - 1)))
    for name, (cnt, sec) in sorted(ACC.items(), key=lambda kv: -kv[1][1]):
        print("   %-24s %5d calls  %8.1f us per frame" % (name, cnt, sec * 1e6 / (nframes - 1)))
    print("   %-24s              %8.1f us per frame" % ("everything else", (total - sum(v[1] for v in ACC.values())) * 1e6 / (nframes - 1)))


if __name__ == "__main__":
    main()
