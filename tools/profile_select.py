#!/usr/bin/env python3
"""rocprofv3 target: N selections of 5000 features on a resident 1080p frame (cfg-2 geometry)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth                      # noqa: E402
from pyfeaturetrack_amd.backend import Context            # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext    # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tc = KLT_TrackingContext()
tc.nPyramidLevels, tc.subsampling = 3, 4
tc.KLTUpdateTCBorder()
ctx = Context(0)
ctx.set_params(params_from_tc(tc))
ctx.upload(0, synth.synth_frame(1920, 1080, 1, 0))
ctx.build_pyramids(0)
ctx.select(0, 5000, use_pyramid=True)
t = time.perf_counter()
for _ in range(n):
    ctx.select_async(0, 1, True, 1, 5000)
ctx.sync()
print("ms per select:", (time.perf_counter() - t) / n * 1e3)
ctx.close()
