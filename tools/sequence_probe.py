#!/usr/bin/env python3
"""Where the host's time goes in KLTTrackSequence: every backend call of the per-frame loop wrapped with a wall-clock timer (1080p, 5000
features, replacement on, 64 frames).  Prints ms per frame and the mean microseconds per frame spent inside each call."""
import collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth, trackSequence                 # noqa: E402
from pyfeaturetrack_amd.backend import Context                      # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext              # noqa: E402

W, H, NF, N = 1920, 1080, 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(cls, name):
    f = getattr(cls, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t
            cnt[name] += 1
    setattr(cls, name, g)


for m in ("track_async", "select_begin", "select_finish", "select_prepare", "build_pyramids", "upload_async", "upload_wait", "upload"):
    wrap(Context, m)
wrap(trackSequence._FrameStager, "next")
base = synth.synth_base(W, H, 3)
frames = [synth.synth_frame(W, H, 3, k, base=base) for k in range(N)]
tc = KLT_TrackingContext()
tc.nPyramidLevels, tc.subsampling = 3, 4
tc.KLTUpdateTCBorder()
tc.sequentialMode = True
trackSequence.KLTTrackSequence(tc, frames[:8], NF)                  # warm-up (allocations)
acc.clear(); cnt.clear()
t = time.perf_counter()
ft = trackSequence.KLTTrackSequence(tc, frames, NF)
el = time.perf_counter() - t
print("%.3f ms per frame over %d frames" % (el / (N - 1) * 1e3, N))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-16s %7.1f us per frame (%d calls)" % (k, v / (N - 1) * 1e6, cnt[k]))
print("  accounted        %7.1f us per frame" % (sum(acc.values()) / (N - 1) * 1e6))
