"""Tracker only, on 1 and 3 streams, for feature lists of different length (the cfg-2 list repeated / truncated): is the time per
launch proportional to the number of features (a throughput bound) or flat (the slowest features' serial chains)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc
tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
p = params_from_tc(tc)
f0, f1 = synth.synth_pair(1920, 1080, seed=1)
for n in (625, 1250, 2500, 5000, 10000, 20000, 40000):
    for nctx in (1, 3):
        ctxs = []
        for c in range(nctx):
            cx = Context(0); cx.set_params(p)
            cx.upload(0, f0); cx.upload(1, f1); cx.build_pyramids(0); cx.build_pyramids(1)
            fl, _ = cx.select(0, 5000, use_pyramid=True)
            big = np.tile(fl, (n + 4999) // 5000)[:n].copy()
            cx.featbuf_upload(0, big); cx.featbuf_upload(1, big)
            ctxs.append(cx)
        def step(i): ctxs[i % nctx].track_async(0, 1, 0, 1, n)
        for i in range(300): step(i)
        for cx in ctxs: cx.sync()
        K = 1200
        t = time.perf_counter()
        for i in range(K): step(i)
        for cx in ctxs: cx.sync()
        dt = (time.perf_counter() - t) / K
        print("%6d features, %d stream(s): %7.2f us per launch = %.2f ns per feature" % (n, nctx, dt * 1e6, dt * 1e9 / n), flush=True)
        for cx in ctxs: cx.close()
