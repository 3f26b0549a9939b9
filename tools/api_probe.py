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


def main():
    w, h, n = 1920, 1080, 5000
    sgf.KLT_verbose = tf.KLT_verbose = 0       # (each module has its own switch, as in the reference)
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    base = synth.synth_base(w, h, 1)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 3.3, -2.1)
    out = {}
    for rep in range(3):
        t = time.perf_counter()
        fl = sgf.KLTSelectGoodFeatures(tc, f0, n)
        out["ms_KLTSelectGoodFeatures"] = (time.perf_counter() - t) * 1e3
        keep = [(f.x, f.y, f.val) for f in fl]
        t = time.perf_counter()
        KLTTrackFeatures(tc, f0, f1, fl)
        out["ms_KLTTrackFeatures"] = (time.perf_counter() - t) * 1e3
        out["tracked"] = sum(1 for f in fl if f.val == 0)
        del keep
    frames = [synth.synth_frame(w, h, 1, k, base=base) for k in range(16)]
    tc2 = KLT_TrackingContext()
    tc2.nPyramidLevels, tc2.subsampling = 3, 4
    tc2.KLTUpdateTCBorder()
    tc2.sequentialMode = True
    for rep in range(2):
        t = time.perf_counter()
        ft = KLTTrackSequence(tc2, frames, n)
        out["ms_per_frame_KLTTrackSequence"] = (time.perf_counter() - t) * 1e3 / (len(frames) - 1)
        tc2.pyramid_last = None
    out["sequence_live_last_row"] = int((ft.val[-1] >= 0).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
