#!/usr/bin/env python3
"""One context, 8 resident cfg-2 pairs, `--batch` pairs per launch, KLT_OPT_L0_STREAM on or off: ms per pair, and -- under
rocprofv3 --kernel-trace -- the timeline of the last passes (tools/trace_timeline.py).  usage: l0_stream_probe.py <0|1> [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth                              # noqa: E402
from pyfeaturetrack_amd.backend import Context                    # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc              # noqa: E402
import bench                                                      # noqa: E402

on = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NP, N = 8, 5000
p = params_from_tc(bench.cfg2_context())
c = Context(0)
c.set_params(p)
for k in range(NP):
    f0, f1 = synth.synth_pair(1920, 1080, seed=k + 1)
    c.upload(2 * k, f0)
    c.upload(2 * k + 1, f1)
    c.build_pyramids(2 * k)
    fl, _ = c.select(2 * k, N, use_pyramid=True)
    c.featbuf_upload(100 + k, fl)
c.set_option(17, on)


def one_pass():
    for g in range(NP // B):
        lps = range(g * B, g * B + B)
        c.build_pyramids_batch([2 * lp + f for lp in lps for f in (0, 1)])
        if B == 1:
            c.track_async(2 * g, 2 * g + 1, 100 + g, 200 + g, N)
        else:
            c.track_batch_async([(2 * lp, 2 * lp + 1, 100 + lp, 200 + lp) for lp in lps], N)


for _ in range(20):
    one_pass()
c.sync()
t = time.perf_counter()
for _ in range(50):
    one_pass()
enq = time.perf_counter() - t
c.sync()
el = time.perf_counter() - t
print("l0_stream %d batch %d: %.4f ms per pair (host enqueue %.4f)" % (on, B, el / (50 * NP) * 1e3, enq / (50 * NP) * 1e3))
c.close()
