"""Asymptotic throughput of the two halves of a cfg-2 step when several contexts (HIP streams) run the SAME half back to back:
pyramid build only, tracker only, both.  Shows how much of a kernel's single-stream time is fixed cost that another stream hides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc

tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
p = params_from_tc(tc)
f0, f1 = synth.synth_pair(1920, 1080, seed=1)
N = 5000
for what in ("pyramids", "tracker", "both"):
    for nctx in (1, 2, 3, 4):
        ctxs = []
        for c in range(nctx):
            cx = Context(0); cx.set_params(p)
            cx.upload(0, f0); cx.upload(1, f1); cx.build_pyramids(0); cx.build_pyramids(1)
            fl, _ = cx.select(0, N, use_pyramid=True)
            cx.featbuf_upload(0, fl); cx.featbuf_upload(1, fl)
            ctxs.append(cx)
        def step(i):
            cx = ctxs[i % nctx]
            if what != "tracker": cx.build_pyramids_batch([0, 1])
            if what != "pyramids": cx.track_async(0, 1, 0, 1, N)
        for i in range(40): step(i)
        for cx in ctxs: cx.sync()
        K = 800
        t = time.perf_counter()
        for i in range(K): step(i)
        for cx in ctxs: cx.sync()
        dt = (time.perf_counter() - t) / K
        print("%-8s %d context(s): %.2f us per step" % (what, nctx, dt * 1e6), flush=True)
        for cx in ctxs: cx.close()
