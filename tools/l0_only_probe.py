import os, sys, time
sys.path.insert(0, "/root/repo")
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc
f0, f1 = synth.synth_pair(1920, 1080, seed=1)
for levels in (1, 2, 3):
    tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = levels, 4; tc.KLTUpdateTCBorder()
    p = params_from_tc(tc)
    for nctx in (1, 2, 3):
        ctxs = []
        for c in range(nctx):
            cx = Context(0); cx.set_params(p); cx.upload(0, f0); cx.upload(1, f1); ctxs.append(cx)
        def step(i): ctxs[i % nctx].build_pyramids_batch([0, 1])
        for i in range(2000): step(i)
        for cx in ctxs: cx.sync()
        K = 3000
        t = time.perf_counter()
        for i in range(K): step(i)
        for cx in ctxs: cx.sync()
        print("levels %d, %d context(s): %.2f us per pyramid pair" % (levels, nctx, (time.perf_counter() - t) / K * 1e6), flush=True)
        for cx in ctxs: cx.close()
