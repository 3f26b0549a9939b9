#!/usr/bin/env python3
"""bench.py's `extra.sequence_from_host` on its own (for rocprofv3 --kernel-trace --memory-copy-trace + tools/trace_copies.py):
    python tools/seq_host_probe.py [4k|1080p] [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "4k"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 128
w, h, n = (3840, 2160, 20000) if size == "4k" else (1920, 1080, 5000)
print(json.dumps(bench.sequence_from_host(0, w, h, n, frames)))

if os.environ.get("KLT_PROBE_HOST_TIMES") == "1":
    import collections
    import time
    from pyfeaturetrack_amd.backend import Context
    acc, cnt = collections.defaultdict(float), collections.Counter()

    def wrap(name):
        f = getattr(Context, name)

        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[name] += time.perf_counter() - t
                cnt[name] += 1
        setattr(Context, name, g)

    for m in ("track_async", "select_begin", "select_finish", "select_prepare", "build_pyramids", "upload_async", "download_wait", "featbuf_download_async"):
        wrap(m)
    r = bench.sequence_from_host(0, w, h, n, frames)
    tot = frames + 2 * 16
    print("instrumented: %.3f ms per frame" % r["ms_per_frame"])
    for k2, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-24s %7.1f us per call (%d calls)" % (k2, v / cnt[k2] * 1e6, cnt[k2]))
