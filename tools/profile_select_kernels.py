#!/usr/bin/env python3
"""Per-kernel-family times (HIP events inside the library) of N selections of 5000 features at 1080p."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth                      # noqa: E402
from pyfeaturetrack_amd.backend import Context            # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext    # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc      # noqa: E402

w, h, nfeat = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 5000)
tc = KLT_TrackingContext()
tc.nPyramidLevels, tc.subsampling = 3, 4
tc.KLTUpdateTCBorder()
ctx = Context(0)
ctx.set_params(params_from_tc(tc))
ctx.upload(0, synth.synth_frame(w, h, 1, 0))
ctx.build_pyramids(0)
ctx.select(0, nfeat, use_pyramid=True)
ctx.timing_enable(True)
for _ in range(10):
    ctx.select_async(0, 1, True, 1, nfeat)
ctx.sync()
print({k["name"]: round(1e3 * k["total_ms"] / k["launches"], 1) for k in ctx.timing_read()})
ctx.close()
