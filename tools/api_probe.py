#!/usr/bin/env python3
"""Latency of the reference-shaped Python API on cfg-2 (1920x1080 pair, 5000 features): what a script written against
PyFeatureTrack sees per call, host list conversion and PCIe included.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyfeaturetrack_amd import selectGoodFeatures as sgf          # noqa: E402
from pyfeaturetrack_amd import synth                               # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext             # noqa: E402
from pyfeaturetrack_amd import trackFeatures as tf                  # noqa: E402
from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures      # noqa: E402
from pyfeaturetrack_amd.trackSequence import KLTTrackSequence      # noqa: E402


def profile():
    """cProfile of 200 KLTTrackFeatures / KLTSelectGoodFeatures calls (the same two frames): where the host time of a call goes"""
    import cProfile
    import pstats
    w, h, n = 1920, 1080, 5000
    sgf.KLT_verbose = tf.KLT_verbose = 0
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    base = synth.synth_base(w, h, 1)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 3.3, -2.1)
    fl = sgf.KLTSelectGoodFeatures(tc, f0, n)
    KLTTrackFeatures(tc, f0, f1, fl)
    for name, fn in (("track", lambda: KLTTrackFeatures(tc, f0, f1, sgf.KLTSelectGoodFeatures(tc, f0, n))),):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            fn()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stderr)
        st.sort_stats("tottime").print_stats(22)


def sequence_figures(w, h, n, seed, nframes, tag):
    """KLTTrackSequence over `nframes` frames held as numpy arrays (16 distinct ones visited up and down, so that consecutive frames
    always differ by one step): ms per frame of the whole call -- first selection, helper thread, table download included."""
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    base = synth.synth_base(w, h, seed)
    distinct = [synth.synth_frame(w, h, seed, k, base=base) for k in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))
    frames = [distinct[order[k % len(order)]] for k in range(nframes)]
    best = None
    for rep in range(3):
        t = time.perf_counter()
        ft = KLTTrackSequence(tc, frames, n)
        ms = (time.perf_counter() - t) * 1e3 / (nframes - 1)
        best = ms if best is None else min(best, ms)
    return {"ms_per_frame_KLTTrackSequence_%s_%d_frames" % (tag, nframes): best,
            "sequence_%s_live_last_row" % tag: int((ft.val[-1] >= 0).sum())}


def main():
    if "--profile" in sys.argv:
        return profile()
    w, h, n = 1920, 1080, 5000
    sgf.KLT_verbose = tf.KLT_verbose = 0       # (each module has its own switch, as in the reference)
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    base = synth.synth_base(w, h, 1)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 3.3, -2.1)
    out = {}
    sel, trk = [], []
    for rep in range(12):
        t = time.perf_counter()
        fl = sgf.KLTSelectGoodFeatures(tc, f0, n)
        sel.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        KLTTrackFeatures(tc, f0, f1, fl)
        trk.append((time.perf_counter() - t) * 1e3)
        out["tracked"] = sum(1 for f in fl if f.val == 0)
    out["ms_KLTSelectGoodFeatures_first_call"], out["ms_KLTTrackFeatures_first_call"] = sel[0], trk[0]
    out["ms_KLTSelectGoodFeatures"] = sorted(sel[2:])[len(sel[2:]) // 2]           # the frames are resident after the first calls
    out["ms_KLTTrackFeatures"] = sorted(trk[2:])[len(trk[2:]) // 2]
    # frames the device has not seen (a fresh array object per call): upload through the pinned ring + pyramids inside the call
    sel, trk = [], []
    for rep in range(8):
        g0, g1 = f0.copy(), f1.copy()
        t = time.perf_counter()
        fl = sgf.KLTSelectGoodFeatures(tc, g0, n)
        sel.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        KLTTrackFeatures(tc, g0, g1, fl)
        trk.append((time.perf_counter() - t) * 1e3)
    out["ms_KLTSelectGoodFeatures_new_frame"] = sorted(sel[1:])[len(sel[1:]) // 2]
    out["ms_KLTTrackFeatures_one_new_frame"] = sorted(trk[1:])[len(trk[1:]) // 2]
    out.update(sequence_figures(w, h, n, 1, 256, "1080p"))
    if "--4k" in sys.argv:
        out.update(sequence_figures(3840, 2160, 20000, 4, 256, "4k"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
