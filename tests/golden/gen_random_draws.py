#!/usr/bin/env python3
"""Golden vectors for RANDOM parameter draws, produced by the reference itself (development container only, like gen_golden.py).

For every draw -- frame size of any parity, pyramid depth, subsampling, window, minimum distance, skipped pixels, pre-smoothing,
residue limit, iteration count, list length -- the reference selects features on frame 0, tracks them into frame 1, and fills the
lost slots from frame 1's candidates (_enforceMinimumDistance with overwriteAllFeatures = False, the level at which the reference
implements replacement: SURVEY a-23), and tracks the resulting list into frame 2 (half the draws in sequential mode).  Inputs are regenerated from the seeds by pyfeaturetrack_amd.synth; the file holds the drawn
parameters and the four feature lists per draw.

    python tests/golden/gen_random_draws.py [--draws 160] [--seed 2026]      ->  tests/golden/random_draws.npz
    python tests/golden/gen_random_draws.py --draws 60 --seed 2027 --max-w 1400 --max-h 1000 --max-n 2000 --out random_draws_large.npz
"""
import argparse
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import build_reference, feats_to_arrays  # noqa: E402
from pyfeaturetrack_amd import synth  # noqa: E402


def draw(rng, max_w=520, max_h=400, max_n=300):
    while True:
        levels = int(rng.integers(1, 4))
        ss = int(rng.choice([2, 4, 8]))
        window = int(rng.choice([3, 5, 7, 9, 11, 15]))
        w = int(rng.integers(60, max_w))
        h = int(rng.integers(60, max_h))
        coarse = ss ** (levels - 1)
        if w // coarse < window + 12 or h // coarse < window + 12:
            continue
        return dict(levels=levels, ss=ss, window=window, w=w, h=h, mindist=int(rng.integers(0, 20)), skip=int(rng.integers(0, 3)),
                    smooth=bool(rng.integers(0, 2)), mr=(None if rng.random() < 0.3 else round(float(rng.uniform(2.0, 30.0)), 3)),
                    n=int(rng.integers(1, max_n)), seed=int(rng.integers(0, 1 << 30)),
                    shift=(round(float(rng.uniform(-2.5, 2.5)), 3), round(float(rng.uniform(-2.5, 2.5)), 3)),
                    min_eig=int(rng.choice([1, 1, 10, 200])), max_iter=int(rng.choice([10, 10, 3, 25])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=160)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--max-w", type=int, default=520)
    ap.add_argument("--max-h", type=int, default=400)
    ap.add_argument("--max-n", type=int, default=300)
    ap.add_argument("--out", default="random_draws.npz")
    a = ap.parse_args()
    refdir = build_reference()
    sys.path.insert(0, refdir)
    os.chdir(refdir)
    warnings.simplefilter("ignore")
    from PIL import Image
    import klt
    import convolve
    import selectGoodFeatures as sgf
    import trackFeatures as tf
    import goodFeaturesUtils as gfu
    sgf.KLT_verbose = 0
    tf.KLT_verbose = 0

    def candidates(tc, pil_img):
        """the sorted candidate list _KLTSelectGoodFeatures builds (selectGoodFeatures.py:187-236)"""
        tmp = np.array(pil_img.convert("F"))
        fimg = convolve.KLTComputeSmoothedImage(tmp, klt.KLTComputeSmoothSigma(tc)) if tc.smoothBeforeSelecting else tmp
        gx, gy = convolve.KLTComputeGradients(fimg, tc.grad_sigma)
        bx, by = tc.borderx, tc.bordery
        hw, hh = tc.window_width / 2, tc.window_height / 2
        bx = hw if bx < hw else bx
        by = hh if by < hh else by
        px, py, pv = gfu.ScanImageForGoodFeatures(gx, gy, bx, by, hw, hh, tc.nSkippedPixels)
        pl = list(zip(pv, px, py))
        pl.sort()
        pl.reverse()
        return pl

    rng = np.random.default_rng(a.seed)
    out, draws = {}, []
    k = skipped = 0
    while k < a.draws:
        t = draw(rng, a.max_w, a.max_h, a.max_n)
        tc = klt.KLT_TrackingContext()
        tc.window_width = tc.window_height = t["window"]
        tc.nPyramidLevels, tc.subsampling = t["levels"], t["ss"]
        tc.KLTUpdateTCBorder()
        tc.mindist, tc.nSkippedPixels, tc.smoothBeforeSelecting = t["mindist"], t["skip"], t["smooth"]
        tc.max_residue, tc.min_eigenvalue, tc.max_iterations = t["mr"], t["min_eig"], t["max_iter"]
        base = synth.synth_base(t["w"], t["h"], t["seed"])
        f0 = synth.shift_frame(base, 0, 0)
        f1 = synth.shift_frame(base, *t["shift"])
        f2 = synth.shift_frame(base, 2 * t["shift"][0], 2 * t["shift"][1])
        pil = [Image.fromarray(f, "L") for f in (f0, f1, f2)]
        tc.sequentialMode = bool(t["seed"] & 1)            # half the draws: the second call reuses the pyramids the first one kept
        try:
            fl = sgf.KLTSelectGoodFeatures(tc, pil[0], t["n"])
        except AttributeError:
            # the reference fails when the candidates run out before the list is full (selectGoodFeatures.py:80 reads .val of a
            # feature that never had one): not a case it defines -- drawn again
            skipped += 1
            continue
        out["d%d_sel_x" % k], out["d%d_sel_y" % k], out["d%d_sel_val" % k] = feats_to_arrays(fl)
        tf.KLTTrackFeatures(tc, pil[0], pil[1], fl)
        out["d%d_trk_x" % k], out["d%d_trk_y" % k], out["d%d_trk_val" % k] = feats_to_arrays(fl)
        sgf._enforceMinimumDistance(candidates(tc, pil[1]), fl, t["w"], t["h"], tc.mindist, tc.min_eigenvalue, False)
        out["d%d_rep_x" % k], out["d%d_rep_y" % k], out["d%d_rep_val" % k] = feats_to_arrays(fl)
        # a second call on the list that now holds tracked (val 0), replaced (val > 0) and lost features; in sequential mode its first
        # image argument is ignored and the pyramids of frame 1 kept by the first call are used (trackFeatures.py:152-161)
        tf.KLTTrackFeatures(tc, pil[1], pil[2], fl)
        out["d%d_trk2_x" % k], out["d%d_trk2_y" % k], out["d%d_trk2_val" % k] = feats_to_arrays(fl)
        draws.append(t)
        print("draw %2d: %s  tracked %d of %d" % (k, t, int((out["d%d_trk_val" % k] == 0).sum()), t["n"]), flush=True)
        k += 1
    print("%d draws the reference could not run were drawn again" % skipped)
    out["draws_json"] = np.frombuffer(json.dumps(draws).encode(), np.uint8)
    for key in list(out):
        if key.endswith(("_x", "_y")):
            assert np.array_equal(out[key], out[key].astype(np.float32)), key      # every position the reference produces is f32-valued
            out[key] = out[key].astype(np.float32)
        elif key.endswith("_val"):
            out[key] = out[key].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, a.out), **out)
    print("wrote", os.path.join(HERE, a.out))


if __name__ == "__main__":
    main()
