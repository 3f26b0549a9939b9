"""What the helper thread of KLTTrackSequence pays per frame: one core copying a pageable 4K / 1080p frame into pinned memory
(klt_host_copy with the thread's serial flag, numpy's own copy, the pool's lanes)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyfeaturetrack_amd._abi import load_library          # noqa: E402
from pyfeaturetrack_amd.backend import Context            # noqa: E402

lib = load_library()
ctx = Context(0)
out = {}
for name, (h, w) in (("4k", (2160, 3840)), ("1080p", (1080, 1920))):
    srcs = [np.random.default_rng(k).integers(0, 255, (h, w), dtype=np.uint8) for k in range(16)]      # 16 distinct pageable frames: nothing stays in a cache
    pins = [ctx.pinned_array((h, w)) for _ in range(4)]

    def best(fn, reps=5):
        b = 1e9
        for _ in range(reps):
            t = time.perf_counter()
            for k in range(32):
                fn(pins[k % 4], srcs[k % 16])
            b = min(b, (time.perf_counter() - t) / 32 * 1e3)
        return b
    lib.klt_host_thread_serial(1)
    serial = best(lambda d, s: lib.klt_host_copy(d.ctypes.data, s.ctypes.data, s.nbytes))
    lib.klt_host_thread_serial(0)
    pooled = best(lambda d, s: lib.klt_host_copy(d.ctypes.data, s.ctypes.data, s.nbytes))
    numpy_copy = best(lambda d, s: np.copyto(d, s))
    out[name] = {"one_core_ms": serial, "pool_lanes_ms": pooled, "numpy_ms": numpy_copy, "lanes": lib.klt_host_lanes(),
                 "one_core_GBps": h * w / serial / 1e6}
ctx.close()
print(json.dumps(out))
