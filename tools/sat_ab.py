import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc
tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
ctx = Context(0); ctx.set_params(params_from_tc(tc))
for (w, h, n) in ((1920, 1080, 5000), (251, 187, 100), (3840, 2160, 20000)):
    if w < 300:
        tc2 = KLT_TrackingContext(); ctx.set_params(params_from_tc(tc2))
    else:
        ctx.set_params(params_from_tc(tc))
    ctx.upload(0, synth.synth_frame(w, h, 1, 0)); ctx.build_pyramids(0)
    res = {}
    for variant in (0, 1):
        ctx.set_option(10, variant)
        fl, placed = ctx.select(0, n, use_pyramid=True)
        val = ctx.select_intermediate(3)
        ctx.timing_enable(True)
        for _ in range(10): ctx.select_async(0, 1, True, 1, n)
        ctx.sync()
        t = {k["name"]: round(1e3*k["total_ms"]/k["launches"],1) for k in ctx.timing_read()}
        ctx.timing_enable(False)
        res[variant] = (fl, val)
        print(w, h, "variant", variant, "placed", placed, {k: t[k] for k in ("sat_rows", "sat_cols")})
    print("   identical eigenvalue maps:", np.array_equal(res[0][1], res[1][1]), " identical lists:", np.array_equal(res[0][0], res[1][0]))
