#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random frame sizes / pyramid geometries / windows, HIP path against the CPU oracle
(pyramids, eigenvalue map, selected list, tracked list, bit for bit).  Not part of the test suite: a longer soak of the kernels'
tile-edge, alignment and fallback paths.   python tests/fuzz/parity_sweep.py [cases] [seed] [max_width] [max_height]
(frames of a megapixel and more take the tall-tile level-0 kernel with the fused first reduction)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyfeaturetrack_amd import synth                       # noqa: E402
from pyfeaturetrack_amd.backend import Context              # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc        # noqa: E402
from helpers import make_tc                                 # noqa: E402
from oracle import klt_oracle as ko                         # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
max_w = int(sys.argv[3]) if len(sys.argv) > 3 else 900
max_h = int(sys.argv[4]) if len(sys.argv) > 4 else 700
ctx = Context(0)
ko.set_threads(8)
bad = 0
t0 = time.time()
for case in range(ncases):
    window = int(rng.choice([3, 5, 7, 7, 7, 9, 15]))
    levels, ss = [(1, 2), (2, 2), (2, 4), (3, 2), (3, 4), (2, 8)][int(rng.integers(6))]
    tc = make_tc(levels=levels, ss=ss, window=window, max_residue=[None, 10.0, 25.0][int(rng.integers(3))])
    need = int(2 * tc.borderx + 40)
    w = int(rng.integers(max(need, 64), max(max_w, need + 65)))
    h = int(rng.integers(max(int(2 * tc.bordery + 40), 48), max(max_h, int(2 * tc.bordery + 40) + 49)))
    if rng.random() < 0.5:
        w -= w % 4                                            # quad-aligned widths take the vector paths
    shift = (float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-2.5, 2.5)))
    base = synth.synth_base(w, h, int(rng.integers(1000)))
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, *shift)
    if rng.random() < 0.3:                                    # flat / saturated regions: zero gradients, ties
        f0[: h // 3, : w // 2] = 255
        f1[: h // 3, : w // 2] = 255
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0); ctx.upload(1, f1)
    ctx.build_pyramids(0); ctx.build_pyramids(1)
    n = int(rng.integers(20, 400)) if w * h < 1000000 or window != 7 else int(rng.integers(2048, 7000))   # long 7x7 lists: the quad tracker
    fl, placed = ctx.select(0, n, use_pyramid=bool(rng.integers(2)))
    val = ctx.select_intermediate(3)
    out, _ = ctx.track(0, 1, fl)
    a0, a1 = f0.astype(np.float32), f1.astype(np.float32)
    P0, P1 = ko.Pyramids(p, a0), ko.Pyramids(p, a1)
    ofl, oval = ko.select_good_features(p, a0, n, want_val=True)
    ok = np.array_equal(val, oval)
    for l in range(levels):
        for pi, wname in enumerate(("img", "gx", "gy")):
            ok &= np.array_equal(ctx.download_level(1, pi, l), P1.level(wname, l))
    sel_ok = all(np.array_equal(fl[k], ofl[k]) for k in ("x", "y", "val"))
    ko.track_features(p, P0, P1, ofl)
    trk_ok = all(np.array_equal(out[k], ofl[k]) for k in ("x", "y", "val"))
    status = "ok" if (ok and sel_ok and trk_ok) else "MISMATCH (images %s, selection %s, tracking %s)" % (ok, sel_ok, trk_ok)
    bad += status != "ok"
    print("%3d: %4dx%-4d window %2d levels %d ss %d n %3d placed %3d tracked %3d  %s" %
          (case, w, h, window, levels, ss, n, placed, int((out["val"] >= 0).sum()), status), flush=True)
print("%d cases, %d mismatches, %.0f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
