#!/usr/bin/env python3
"""cfg-2 pairs per second for S contexts (HIP streams) x B pairs per batched launch: is a batch of pairs per launch better than more pairs in
flight?  Every context holds B resident 1080p pairs; a step = one batched pyramid build of its 2B frames + one tracker launch for its B pairs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth                                  # noqa: E402
from pyfeaturetrack_amd.backend import Context                        # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc                  # noqa: E402

W, H, NF = 1920, 1080, 5000
tc = KLT_TrackingContext()
tc.nPyramidLevels, tc.subsampling = 3, 4
tc.KLTUpdateTCBorder()
p = params_from_tc(tc)
f0, f1 = synth.synth_pair(W, H, seed=1)
for S, B in [(3, 1), (1, 1), (2, 2), (3, 2), (1, 4), (2, 3), (2, 4), (1, 8)]:
    ctxs = []
    for s in range(S):
        c = Context(0)
        c.set_params(p)
        for b in range(B):
            c.upload(2 * b, f0)
            c.upload(2 * b + 1, f1)
        c.build_pyramids_batch(list(range(2 * B)), sync=True)
        c.featbuf_alloc(900, B * NF)
        c.featbuf_alloc(901, B * NF)
        for b in range(B):
            c.featbuf_view(910 + b, 900, b * NF, NF)
            c.featbuf_view(930 + b, 901, b * NF, NF)
            c.select_async(2 * b, 1, True, 910 + b, NF)
        c.sync()
        ctxs.append(c)
    table = [(2 * b, 2 * b + 1, 910 + b, 930 + b) for b in range(B)]
    slots = list(range(2 * B))

    def run(n):
        for i in range(n):
            c = ctxs[i % S]
            c.build_pyramids_batch(slots)
            c.track_batch_async(table, NF)
        for c in ctxs:
            c.sync()
    run(60)
    best = 1e9
    for _ in range(5):
        n = 600 // B
        t = time.perf_counter()
        run(n)
        best = min(best, (time.perf_counter() - t) / (n * B))
    print("streams %d x batch %d: %.4f ms per pair" % (S, B, best * 1e3), flush=True)
    for c in ctxs:
        c.close()
