#!/usr/bin/env python3
"""Where does the time of the pipelined-ingest loop go?  (host enqueue vs device)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc

tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
ctx = Context(0); ctx.set_params(params_from_tc(tc))
f0, f1 = synth.synth_pair(1920, 1080, 1)
pins = {s: ctx.pinned_array((1080, 1920)) for s in range(4)}
for s in (0, 2):
    pins[s][:] = f0; pins[s + 1][:] = f1
    ctx.upload(s, f0); ctx.upload(s + 1, f1)
ctx.build_pyramids(0)
fl, _ = ctx.select(0, 5000, use_pyramid=True)
ctx.featbuf_upload(0, fl); ctx.featbuf_upload(1, fl)
N = 64
def run(name, body):
    ctx.sync(); t = time.perf_counter()
    for i in range(N): body(i)
    te = time.perf_counter() - t
    ctx.sync(); tt = time.perf_counter() - t
    print("%-46s enqueue %.1f us/step   total %.1f us/step" % (name, te / N * 1e6, tt / N * 1e6))
def compute(i):
    a = 0 if i % 2 == 0 else 2
    ctx.build_pyramids_batch([a, a + 1]); ctx.track_async(a, a + 1, 0, 1, 5000)
def up_only(i):
    a = 0 if i % 2 == 0 else 2
    ctx.upload_async(a, pins[a]); ctx.upload_async(a + 1, pins[a + 1])
def both(i):
    up_only(i); compute(i)
def sync_up(i):
    a = 0 if i % 2 == 0 else 2
    ctx.upload(a, f0); ctx.upload(a + 1, f1); compute(i)
for _ in range(2):
    run("compute only", compute)
    run("async uploads only", up_only)
    run("async uploads + compute", both)
    run("synchronous uploads (pageable) + compute", sync_up)
ctx.close()

# ---- which call blocks?
import collections
ctx = Context(0); ctx.set_params(params_from_tc(tc))
pins = {s: ctx.pinned_array((1080, 1920)) for s in range(4)}
for s in (0, 2):
    pins[s][:] = f0; pins[s + 1][:] = f1
    ctx.upload(s, f0); ctx.upload(s + 1, f1)
ctx.build_pyramids(0)
ctx.featbuf_upload(0, fl); ctx.featbuf_upload(1, fl)
acc = collections.defaultdict(float)
def timed(name, fn, *a):
    t = time.perf_counter(); fn(*a); acc[name] += time.perf_counter() - t
ctx.sync()
for i in range(N):
    a = 0 if i % 2 == 0 else 2
    timed("upload_async A", ctx.upload_async, a, pins[a])
    timed("upload_async B", ctx.upload_async, a + 1, pins[a + 1])
    timed("build_pyramids_batch", ctx.build_pyramids_batch, [a, a + 1])
    timed("track_async", ctx.track_async, a, a + 1, 0, 1, 5000)
ctx.sync()
for k, v in acc.items():
    print("%-22s %.1f us per call" % (k, v / N * 1e6))
ctx.close()
