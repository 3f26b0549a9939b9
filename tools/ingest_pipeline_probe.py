#!/usr/bin/env python3
"""The pipelined-ingest loop of bench.py (`extra.pcie_pipelined_*`) on its own: pairs of 1080p frames from pinned host memory, uploads of
pair i + 1 issued before the kernels of pair i, records into a 16-row device table read back every 16 pairs.  Run it under
`rocprofv3 --kernel-trace --memory-copy-trace` and feed the directory to tools/trace_copies.py to see what the link and the GPU do.

    python tools/ingest_pipeline_probe.py [pairs in rotation] [steps] [uploads ahead]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth                                     # noqa: E402
from pyfeaturetrack_amd.backend import Context                           # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                   # noqa: E402

NPIN = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 128
AHEAD = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W, H, NF, NT = 1920, 1080, 5000, 16
tc = KLT_TrackingContext()
tc.nPyramidLevels, tc.subsampling = 3, 4
tc.KLTUpdateTCBorder()
ctx = Context(0)
ctx.configure(tc)
pins = []
for lp in range(NPIN):
    f0, f1 = synth.synth_pair(W, H, seed=lp + 1)
    a, b = ctx.pinned_array((H, W)), ctx.pinned_array((H, W))
    a[:], b[:] = f0, f1
    pins.append((a, b))
    ctx.upload(2 * lp, f0)
    ctx.upload(2 * lp + 1, f1)
    ctx.build_pyramids(2 * lp)
    ctx.select_async(2 * lp, 1, True, 500 + lp, NF)
SYNC_DOWNLOAD = os.environ.get("KLT_PROBE_SYNC_DOWNLOAD") == "1"      # round 3's loop: a synchronous read-back every 16 pairs
TAB = 100
ctx.featbuf_alloc(TAB, 2 * NT * NF)
for k in range(2 * NT):
    ctx.featbuf_view(TAB + 1 + k, TAB, k * NF, NF)
HALVES = (300, 301)
for hlf in range(2):
    ctx.featbuf_view(HALVES[hlf], TAB, hlf * NT * NF, NT * NF)
from pyfeaturetrack_amd.backend import FEAT_DTYPE  # noqa: E402
host_tab = [ctx.pinned_array((NT * NF,), FEAT_DTYPE) for _ in range(2)]


def send(i):
    lp = i % NPIN
    ctx.upload_async(2 * lp, pins[lp][0])
    ctx.upload_async(2 * lp + 1, pins[lp][1])


def step(i, host):
    lp = i % NPIN
    t = time.perf_counter()
    send(i + AHEAD)
    host[0] += time.perf_counter() - t
    t = time.perf_counter()
    ctx.build_pyramids_batch([2 * lp, 2 * lp + 1])
    ctx.track_async(2 * lp, 2 * lp + 1, 500 + lp, TAB + 1 + i % (2 * NT), NF)
    host[1] += time.perf_counter() - t
    if i % NT == NT - 1:
        t = time.perf_counter()
        win = i // NT
        if SYNC_DOWNLOAD:
            ctx.featbuf_download(HALVES[win % 2], NT * NF)
        else:
            ctx.download_wait()                                  # the previous window's records
            ctx.featbuf_download_async(HALVES[win % 2], host_tab[win % 2])
        host[2] += time.perf_counter() - t


for j in range(AHEAD):
    send(j)
host = [0.0, 0.0, 0.0]
for i in range(NT):
    step(i, host)
ctx.sync()
host = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for i in range(NT, NT + STEPS):
    step(i, host)
ctx.download_wait()
ctx.sync()
el = (time.perf_counter() - t0) / STEPS
print("%d pairs in rotation, uploads %d ahead: %.1f us per pair = %.1f GB/s; host per pair: uploads %.1f us, build + track %.1f us, table read-back %.1f us"
      % (NPIN, AHEAD, el * 1e6, 2 * W * H / el / 1e9, host[0] / STEPS * 1e6, host[1] / STEPS * 1e6, host[2] / STEPS * 1e6))
ctx.close()
