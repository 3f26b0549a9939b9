#!/usr/bin/env python3
"""Golden vectors for the reference's module-level helper functions, produced by RUNNING the reference (development container only,
like gen_golden.py, whose build step this script reuses): computeIntensityDifference / computeGradientSum
(trackFeaturesUtils.pyx:90-97, :130-142), _trackFeature (trackFeatures.py:67-136), _enforceMinimumDistance and _fillFeaturemap
(selectGoodFeatures.py:18-25, :45-135).  Writes tests/golden/literal_boundary.npz: inputs and expected outputs, data only.

    python tests/golden/gen_literal_boundary.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import build_reference  # noqa: E402


def main():
    refdir = build_reference()
    sys.path.insert(0, refdir)
    os.chdir(refdir)
    warnings.simplefilter("ignore")
    from PIL import Image
    import klt
    import selectGoodFeatures as sgf
    import trackFeatures as tf
    import trackFeaturesUtils as tfu

    sgf.KLT_verbose = tf.KLT_verbose = 0
    out = {}

    # ------------------------------------------------------------ computeIntensityDifference / computeGradientSum
    rng = np.random.default_rng(21)
    img = (rng.random((60, 72)) * 255).astype(np.float32)
    out["cid_img"] = img
    for w, cnt in ((7, 40), (15, 20)):
        xs = (12 + rng.random(cnt) * 44).astype(np.float32)
        ys = (12 + rng.random(cnt) * 32).astype(np.float32)
        xs[:3] = np.floor(xs[:3])
        p1 = (rng.random((cnt, w, w)) * 255).astype(np.float32)
        diffs, works, sums = [], [], []
        for k in range(cnt):
            work = np.full((w, w), -7.0, np.float32)
            d = np.zeros(w * w, np.float32)
            r = tfu.computeIntensityDifference(p1[k], img, float(xs[k]), float(ys[k]), work, d)
            assert r is None
            diffs.append(d)
            works.append(work.copy())
            g = np.zeros((w * w, 2), np.float32)
            work2 = np.empty((w, w), np.float32)
            tfu.computeGradientSum(p1[k], img, float(xs[k]), float(ys[k]), work2, g, 0)
            tfu.computeGradientSum(p1[(k + 1) % cnt], img, float(ys[k]), float(xs[k]) * 0.5 + 8, work2, g, 1)
            sums.append(g)
        out["cid_x_%d" % w], out["cid_y_%d" % w], out["cid_p1_%d" % w] = xs, ys, p1
        out["cid_diff_%d" % w], out["cid_work_%d" % w], out["cgs_sum_%d" % w] = np.stack(diffs), np.stack(works), np.stack(sums)

    # ------------------------------------------------------------ _trackFeature: every call of one KLTTrackFeatures run on img0 -> img1
    img0 = Image.open(os.path.join(HERE, "img0.pgm"))
    img1 = Image.open(os.path.join(HERE, "img1.pgm"))
    for tag, mr, retain in (("r10", 10.0, False), ("rnone", None, False), ("retain", 10.0, True)):
        tc = klt.KLT_TrackingContext()
        tc.max_residue = mr
        tc.retainTrackers = retain
        fl = sgf.KLTSelectGoodFeatures(tc, img0, 100)
        rows = []
        inner = tf._trackFeature

        def recorder(x1, y1, x2, y2, i1, gx1, gy1, i2, gx2, gy2, tc_, rows=rows, inner=inner):
            r = inner(x1, y1, x2, y2, i1, gx1, gy1, i2, gx2, gy2, tc_)
            rows.append((float(x1), float(y1), float(x2), float(y2), i1.shape[1], float(r[0]), float(r[1]), float(r[2])))
            return r
        tf._trackFeature = recorder
        try:
            tf.KLTTrackFeatures(tc, img0, img1, fl)
        finally:
            tf._trackFeature = inner
        out["tf_%s" % tag] = np.array(rows, np.float64)

    # ------------------------------------------------------------ _enforceMinimumDistance on point lists nobody sorted
    class F(object):
        pass
    ncols, nrows = 200, 150
    cases = []
    for ci, (npts, nfeat, mindist, min_eig, overwrite, nlive) in enumerate([
            (300, 40, 10, 1, True, 0), (300, 40, 10, 1, False, 12), (500, 60, 1, 0.2, True, 0), (500, 60, 0, 50, False, 20),
            (80, 50, 5, 1, True, 7), (80, 50, 5, 1, False, 50), (400, 30, 25, 300, False, 5), (10, 30, 3, 1, True, 30)]):
        r = np.random.default_rng(100 + ci)
        px = r.integers(0, ncols, npts)
        py = r.integers(0, nrows, npts)
        pv = (r.random(npts) * 1000).astype(np.float32)
        pv[r.random(npts) < 0.15] = 0.5
        if ci % 2 == 0:
            o = np.argsort(-pv, kind="stable")
            px, py, pv = px[o], py[o], pv[o]
        px[5:8], py[5:8] = px[4], py[4]                                   # duplicates of one position
        pointlist = [(float(v), int(x), int(y)) for v, x, y in zip(pv, px, py)]
        fl = []
        live = set(r.choice(nfeat, nlive, replace=False).tolist()) if nlive else set()
        for i in range(nfeat):
            f = F()
            if i in live:
                f.x, f.y, f.val = float(r.uniform(0, ncols - 1)), float(r.uniform(0, nrows - 1)), int(r.integers(0, 500))
            else:
                f.x, f.y, f.val = -1, -1, -1
            fl.append(f)
        fin = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
        sgf._enforceMinimumDistance(pointlist, fl, ncols, nrows, mindist, min_eig, overwrite)
        fout = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
        out["emd_%d_points" % ci] = np.array([(v, x, y) for v, x, y in pointlist], np.float64)
        out["emd_%d_in" % ci], out["emd_%d_out" % ci] = fin, fout
        cases.append((ncols, nrows, mindist, min_eig, int(overwrite)))
    out["emd_cases"] = np.array(cases, np.float64)

    # ------------------------------------------------------------ _fillFeaturemap
    fm_cases = [(3, 4, 2, 12, 9), (0, 0, 3, 12, 9), (11, 8, 5, 12, 9), (6, 4, 0, 12, 9), (6, 4, -1, 12, 9)]
    maps = []
    for x, y, md, nc_, nr_ in fm_cases:
        fm = [False] * (nc_ * nr_)
        got = sgf._fillFeaturemap(x, y, fm, md, nc_, nr_)
        assert got is fm
        maps.append(np.array(fm, bool))
    out["ffm_cases"], out["ffm_maps"] = np.array(fm_cases, np.int64), np.stack(maps)

    np.savez_compressed(os.path.join(HERE, "literal_boundary.npz"), **out)
    print("wrote literal_boundary.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
