"""Throughput of independent cfg-2 pairs when alternate pairs go to 1, 2 or 3 contexts (each its own HIP stream, no
events between them): kernels of different pairs overlap on the GPU."""
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
N = 5000
for nctx in (1, 2, 3, 4):
    ctxs = []
    for c in range(nctx):
        ctx = Context(0); ctx.set_params(p)
        ctx.upload(0, f0); ctx.upload(1, f1)
        ctx.build_pyramids(0)
        fl, placed = ctx.select(0, N, use_pyramid=True)
        ctx.featbuf_upload(0, fl); ctx.featbuf_upload(1, fl)
        ctxs.append(ctx)
    def step(i):
        c = ctxs[i % nctx]
        c.build_pyramids_batch([0, 1]); c.track_async(0, 1, 0, 1, N)
    for i in range(40): step(i)
    for c in ctxs: c.sync()
    K = 600
    t = time.perf_counter()
    for i in range(K): step(i)
    for c in ctxs: c.sync()
    dt = (time.perf_counter() - t) / K
    out = [c.featbuf_download(1, N) for c in ctxs]
    same = all(np.array_equal(o["x"], out[0]["x"]) and np.array_equal(o["val"], out[0]["val"]) for o in out)
    print("contexts %d: %.4f ms per pair, %.1f M features/s, identical results across contexts: %s" % (nctx, dt * 1e3, N / dt / 1e6, same))
    for c in ctxs: c.close()
