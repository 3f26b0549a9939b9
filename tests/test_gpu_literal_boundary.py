"""The reference's own function names on the device (SURVEY 8 b; DESIGN section 2 "the Cython def layer itself" and "module-level
helpers"): ScanImageForGoodFeatures, extractImagePatchSlow, trackFeatureIterateCKLT, computeIntensityDifference, computeGradientSum
(setup.py:8-9), _trackFeature (trackFeatures.py:67-136), _enforceMinimumDistance (selectGoodFeatures.py:45-135), and the ctypes stub of
INTEGRATION.md section B run as printed -- against vectors recorded from the reference (tests/golden/literal_boundary.npz, cfg1.npz,
patches.npz).  `_convolveSeparate` is in test_gpu_convolve.py.  (Folded by component from the round-4 / 5 files in round 6: the tests are unchanged.)"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import REPO
from helpers import _api_modules, _records, default_cache, default_lists, make_tc, params_from_tc
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.params import affine_params_from_tc

pytestmark = pytest.mark.gpu


def _compat(name):
    """module `name` as a script written against the reference imports it: the compat directory on sys.path"""
    import importlib
    d = os.path.join(REPO, "pyfeaturetrack_amd", "compat")
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module(name)


def test_scan_image_for_good_features_is_the_reference_function(cfg1):
    """goodFeaturesUtils.ScanImageForGoodFeatures (goodFeaturesUtils.pyx:35-73) under its own module and function name: the
    eigenvalue of every candidate window of img0's gradient images equals the reference's list (tests/golden/cfg1.npz `sel_val`, written
    by gen_golden.py from the reference's own call, with the Python floats 30.0 / 3.5 it passes for the borders and half-windows), the
    coordinate lists are the reference's, and the elements have the reference's types."""
    gfu = _compat("goodFeaturesUtils")
    import pyfeaturetrack_amd.goodFeaturesUtils as real
    assert gfu is real
    px, py, pv = gfu.ScanImageForGoodFeatures(cfg1["sel_gx"], cfg1["sel_gy"], 30.0, 30.0, 3.5, 3.5, 0)
    want = cfg1["sel_val"]
    ny, nx = want.shape
    assert len(px) == len(py) == len(pv) == nx * ny and isinstance(pv[0], float) and isinstance(px[0], np.int32)
    assert np.array_equal(np.array(pv, np.float32).reshape(ny, nx), want)
    assert np.array_equal(np.array(px).reshape(ny, nx), np.tile(np.arange(30, 290, dtype=np.int32), (ny, 1)))
    assert np.array_equal(np.array(py).reshape(ny, nx), np.repeat(np.arange(30, 210, dtype=np.int32), nx).reshape(ny, nx))
    # the sorted head the reference's selection walks (selectGoodFeatures.py:234-236) follows from these three lists alone
    pl = sorted(zip(pv, px, py), reverse=True)[:2000]
    assert np.array_equal(np.array([p[0] for p in pl], np.float32), cfg1["sel_sorted_val"][:2000])
    assert np.array_equal(np.array([p[1] for p in pl]), cfg1["sel_sorted_x"][:2000])
    # skipped pixels: every third candidate of the same map
    px3, py3, pv3 = gfu.ScanImageForGoodFeatures(cfg1["sel_gx"], cfg1["sel_gy"], 30, 30, 3, 3, 2)
    assert np.array_equal(np.array(pv3, np.float32).reshape(len(range(30, 210, 3)), -1), want[::3, ::3])
    with pytest.raises(Exception):
        gfu.ScanImageForGoodFeatures(cfg1["sel_gx"], cfg1["sel_gy"], 2, 2, 3, 3, 0)          # the reference reads outside the tables here


def test_extract_image_patch_slow_is_the_reference_function(golden_dir):
    """trackFeaturesUtils.extractImagePatchSlow (trackFeaturesUtils.pyx:14-51): 300 7x7 and 100 15x15 patches at random sub-pixel
    positions equal the reference's (tests/golden/patches.npz, written from the reference's own function)."""
    tfu = _compat("trackFeaturesUtils")
    g = np.load(os.path.join(golden_dir, "patches.npz"))
    for w in (7, 15):
        for k in range(len(g["x_%d" % w])):
            got = tfu.extractImagePatchSlow(g["img"], g["x_%d" % w][k], g["y_%d" % w][k], w, w)
            assert got.dtype == np.float32 and got.shape == (w, w)
            assert np.array_equal(got, g["patch_%d" % w][k]), (w, k)
    with pytest.raises(AssertionError):
        tfu.extractImagePatchSlow(g["img"], 2.5, 20.0, 7, 7)                               # the footprint leaves the image (:35)


@pytest.mark.parametrize("tag", ["r10", "rnone"])
def test_track_feature_iterate_is_the_reference_function(cfg1, tag):
    """trackFeaturesUtils.trackFeatureIterateCKLT (trackFeaturesUtils.pyx:393-459): every one of the 200 calls the reference made while
    tracking 100 features img0 -> img1 (recorded by gen_golden.py: position in, level, position out, status, iterations) is repeated
    through the compat module -- template patches from extractImagePatchSlow on the reference's own pyramid planes -- and returns
    exactly what the reference returned."""
    tfu = _compat("trackFeaturesUtils")
    tc = make_tc(max_residue=10.0 if tag == "r10" else None)
    rows = cfg1["trk100_%s_iter" % tag]
    feat = -1
    for row in rows:
        x2, y2, width, x2o, y2o, status, iters = row
        level = 1 if int(width) == 80 else 0
        if level == 1:
            feat += 1
        x1 = np.float32(cfg1["sel100_x"][feat]) / np.float32(4 ** level)
        y1 = np.float32(cfg1["sel100_y"][feat]) / np.float32(4 ** level)
        planes = [cfg1["p0_%s_%d" % (n, level)] for n in ("gx", "gy", "img")]
        gxp, gyp, ip = (tfu.extractImagePatchSlow(p, x1, y1, 7, 7) for p in planes)
        got = tfu.trackFeatureIterateCKLT(x2, y2, gxp, gyp, ip, cfg1["p1_img_%d" % level], cfg1["p1_gx_%d" % level],
                                          cfg1["p1_gy_%d" % level], tc)
        assert got == (x2o, y2o, int(status), int(iters)), (feat, level, got, tuple(row))
    assert feat == 99


def test_the_integration_stub_runs_as_printed(tmp_path, golden_dir, cfg1):
    """INTEGRATION.md section B calls its ctypes stub "complete, runnable": the code block is cut out of the document, saved as
    klt_gpu_binding.py and driven in a fresh process the way a reference maintainer would (the reference's module names `klt`,
    `convolve`, `klt_util` resolved through compat/, libkltgpu.so found through LD_LIBRARY_PATH) -- 100 features selected on img0 and
    tracked into img1 equal the reference's own lists (tests/golden/cfg1.npz)."""
    pytest.importorskip("PIL.Image")
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    section = text[text.index("## B."):]
    m = re.search(r"```python\n(.*?)```", section, re.S)
    assert m and "klt_gpu_binding.py" in m.group(1)
    (tmp_path / "klt_gpu_binding.py").write_text(m.group(1))
    (tmp_path / "drive.py").write_text('''
import numpy as np
from PIL import Image
from klt import KLT_TrackingContext
import klt_gpu_binding as b
tc = KLT_TrackingContext()
tc.max_residue = 10.0
ctx = b.open_context(tc)
b.upload(ctx, 0, Image.open("img0.pgm"))
b.upload(ctx, 1, Image.open("img1.pgm"))
fl = b.select(ctx, 0, 100)
sel = [(f.x, f.y, f.val) for f in fl]
k = b.track(ctx, 0, 1, fl)
trk = [(f.x, f.y, f.val) for f in fl]
np.savez("out.npz", sel=np.array(sel, np.float64), trk=np.array(trk, np.float64), k=k)
''')
    import shutil
    for name in ("img0.pgm", "img1.pgm"):
        shutil.copy(os.path.join(golden_dir, name), tmp_path / name)
    csrc = os.path.join(REPO, "pyfeaturetrack_amd", "csrc")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, "pyfeaturetrack_amd", "compat")]),
               LD_LIBRARY_PATH=os.pathsep.join([csrc, os.environ.get("LD_LIBRARY_PATH", "")]))
    r = subprocess.run([sys.executable, "drive.py"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = np.load(tmp_path / "out.npz")
    assert np.array_equal(out["sel"][:, 0], cfg1["sel100_x"]) and np.array_equal(out["sel"][:, 1], cfg1["sel100_y"])
    assert np.array_equal(out["sel"][:, 2].astype(np.int64), cfg1["sel100_val"])
    want_val = cfg1["trk100_r10_val"]
    assert np.array_equal(out["trk"][:, 2].astype(np.int64), want_val) and int(out["k"]) == int((want_val >= 0).sum())
    ok = want_val >= 0
    assert np.array_equal(out["trk"][ok, 0], cfg1["trk100_r10_x"][ok]) and np.array_equal(out["trk"][ok, 1], cfg1["trk100_r10_y"][ok])


@pytest.fixture(scope="module")
def literal(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "literal_boundary.npz"))


def test_compute_intensity_difference_and_gradient_sum(literal):
    """compat/trackFeaturesUtils.computeIntensityDifference / computeGradientSum (trackFeaturesUtils.pyx:90-97, :130-142) against what the
    reference's functions wrote into `out` and `workingPatch` for 40 + 20 windows of 7x7 / 15x15 (tests/golden/gen_literal_boundary.py)."""
    from pyfeaturetrack_amd import trackFeaturesUtils as tfu
    img = literal["cid_img"]
    for w in (7, 15):
        xs, ys, p1 = literal["cid_x_%d" % w], literal["cid_y_%d" % w], literal["cid_p1_%d" % w]
        cnt = len(xs)
        for k in range(cnt):
            work = np.full((w, w), -7.0, np.float32)
            d = np.zeros(w * w, np.float32)
            assert tfu.computeIntensityDifference(p1[k], img, float(xs[k]), float(ys[k]), work, d) is None
            assert np.array_equal(work, literal["cid_work_%d" % w][k]) and np.array_equal(d, literal["cid_diff_%d" % w][k])
            g = np.zeros((w * w, 2), np.float32)
            work2 = np.empty((w, w), np.float32)
            tfu.computeGradientSum(p1[k], img, float(xs[k]), float(ys[k]), work2, g, 0)
            tfu.computeGradientSum(p1[(k + 1) % cnt], img, float(ys[k]), float(xs[k]) * 0.5 + 8, work2, g, 1)
            assert np.array_equal(g, literal["cgs_sum_%d" % w][k])
    with pytest.raises(ValueError):
        tfu.computeIntensityDifference(p1[0].astype(np.float64), img, 20.0, 20.0, np.empty((15, 15), np.float32), np.zeros(225, np.float32))


@pytest.mark.parametrize("tag,mr,retain", [("r10", 10.0, False), ("rnone", None, False), ("retain", 10.0, True)])
def test_track_feature_under_the_reference_name(literal, golden_dir, tag, mr, retain):
    """trackFeatures._trackFeature (trackFeatures.py:67-136) on every call the reference made while tracking img0 -> img1 (100 features x
    2 levels, three tracking contexts): same (status, x2, y2), Python floats equal bit for bit."""
    import os
    from conftest import read_pgm
    sgf, trk = _api_modules()
    img0, img1 = read_pgm(os.path.join(golden_dir, "img0.pgm")), read_pgm(os.path.join(golden_dir, "img1.pgm"))
    tc = make_tc(max_residue=mr, retainTrackers=retain)
    p1, p1x, p1y, p2, p2x, p2y = trk.ComputeImagePyramids(tc, img0, img1)
    level = {320: 0, 80: 1}
    rows = literal["tf_%s" % tag]
    assert len(rows) == 200
    statuses = set()
    for x1, y1, x2, y2, nc, st, xo, yo in rows:
        r = level[int(nc)]
        got = trk._trackFeature(x1, y1, x2, y2, p1.img[r], p1x.img[r], p1y.img[r], p2.img[r], p2x.img[r], p2y.img[r], tc)
        assert (got[0], float(got[1]), float(got[2])) == (int(st), xo, yo), (x1, y1, r)
        statuses.add(int(st))
    assert (statuses == {0}) if (retain or mr is None) else (len(statuses) >= 2)


def test_enforce_minimum_distance_under_the_reference_name(literal):
    """selectGoodFeatures._enforceMinimumDistance / _fillFeaturemap (selectGoodFeatures.py:18-25, :45-135) called directly, on point
    lists nobody sorted (duplicates, values below the threshold), with and without live features to keep, mindist 0 .. 25, both
    list kinds (a list this package made; plain objects): the reference's lists."""
    from pyfeaturetrack_amd.klt import KLT_Feature, new_feature_list
    sgf, _ = _api_modules()
    for ci, (ncols, nrows, mindist, min_eig, overwrite) in enumerate(literal["emd_cases"]):
        points = [(float(v), int(x), int(y)) for v, x, y in literal["emd_%d_points" % ci]]
        fin, want = literal["emd_%d_in" % ci], literal["emd_%d_out" % ci]
        for kind in ("package list", "plain objects"):
            fl = new_feature_list(len(fin)) if kind == "package list" else [KLT_Feature() for _ in fin]
            for f, (x, y, v) in zip(fl, fin):
                if v >= 0:
                    f.x, f.y, f.val = float(x), float(y), int(v)
            got = sgf._enforceMinimumDistance(points, fl, int(ncols), int(nrows), int(mindist), float(min_eig) if min_eig != int(min_eig) else int(min_eig),
                                              bool(overwrite))
            assert got is fl
            have = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
            assert np.array_equal(have, want), "case %d (%s)" % (ci, kind)
            placed = [f for f, before, after in zip(fl, fin, want) if after[2] > 0 and tuple(before) != tuple(after)]
            assert all(type(f.x) is int and type(f.y) is int for f in placed)     # newly placed: Python ints (:116-119)
    for (x, y, md, nc_, nr_), want in zip(literal["ffm_cases"], literal["ffm_maps"]):
        fm = [False] * int(nc_ * nr_)
        assert sgf._fillFeaturemap(int(x), int(y), fm, int(md), int(nc_), int(nr_)) is fm
        assert np.array_equal(np.array(fm, bool), want)


def test_replacement_through_enforce_minimum_distance_by_name(cfg1, img1):
    """SURVEY a-23's pin, literally: the reference's replacement = _enforceMinimumDistance(sorted pointlist, featurelist, ...,
    overwriteAllFeatures=False) on img1's candidates -- here the candidates come from the compat ScanImageForGoodFeatures on the device's
    gradients, are sorted as the reference sorts them (selectGoodFeatures.py:234-236) and go through the function under its own name;
    the list equals the reference's `repl_out_*`."""
    from pyfeaturetrack_amd import convolve, goodFeaturesUtils
    from pyfeaturetrack_amd.klt import new_feature_list
    from pyfeaturetrack_amd.klt_util import KLTComputeSmoothSigma
    sgf, _ = _api_modules()
    tc = make_tc(max_residue=10.0)
    smooth = convolve.KLTComputeSmoothedImage(img1.astype(np.float32), KLTComputeSmoothSigma(tc))
    gx, gy = convolve.KLTComputeGradients(smooth, tc.grad_sigma)
    px, py, pv = goodFeaturesUtils.ScanImageForGoodFeatures(gx, gy, tc.borderx, tc.bordery, tc.window_width / 2, tc.window_height / 2,
                                                            tc.nSkippedPixels)
    pointlist = sorted(zip(pv, px, py), reverse=True)
    fl = new_feature_list(100)
    for f, x, y, v in zip(fl, cfg1["repl_in_x"], cfg1["repl_in_y"], cfg1["repl_in_val"]):
        f.x, f.y, f.val = float(x), float(y), int(v)
    sgf._enforceMinimumDistance(pointlist, fl, 320, 240, tc.mindist, tc.min_eigenvalue, False)
    have = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
    assert np.array_equal(have[:, 0], cfg1["repl_out_x"]) and np.array_equal(have[:, 1], cfg1["repl_out_y"])
    assert np.array_equal(have[:, 2], cfg1["repl_out_val"].astype(np.float64))


def test_enforce_minimum_distance_on_random_point_lists_vs_the_checker():
    """selectGoodFeatures._enforceMinimumDistance (klt_min_distance_walk) against oracle/min_distance_walk.py -- the plain-Python
    restatement pinned to the reference's own outputs -- on 60 random cases: frames from 64 x 48 to 2400 x 1800 (with a small
    minimum distance the walk's occupancy grid no longer fits in LDS: the global-memory grid), 0 .. 6000 points in any order with
    duplicates and sub-threshold values, lists of 1 .. 900 features with any share alive, minimum distances 0 .. 40, both modes."""
    from oracle.min_distance_walk import enforce_minimum_distance
    from pyfeaturetrack_amd.klt import new_feature_list
    sgf, _ = _api_modules()
    rng = np.random.default_rng(2026)
    big = 0
    for case in range(60):
        ncols, nrows = (int(rng.integers(64, 400)), int(rng.integers(48, 300))) if case % 3 else (int(rng.integers(1500, 2400)), int(rng.integers(1200, 1800)))
        mindist = int(rng.choice([0, 1, 2, 3, 5, 10, 17, 40]))
        npts = int(rng.integers(0, 6000 if case % 3 == 0 else 1500))
        nfeat = int(rng.integers(1, 900 if case % 3 == 0 else 200))
        overwrite = bool(rng.integers(0, 2))
        min_eig = float(rng.choice([0.2, 1, 1, 30, 400]))
        px, py = rng.integers(0, ncols, npts), rng.integers(0, nrows, npts)
        pv = (rng.random(npts) * 1000).astype(np.float32)
        pv[rng.random(npts) < 0.1] = 0.5
        if npts > 8:
            px[3:6], py[3:6] = px[2], py[2]
        if rng.integers(0, 2):
            o = np.argsort(-pv, kind="stable")
            px, py, pv = px[o], py[o], pv[o]
        points = [(float(v), int(x), int(y)) for v, x, y in zip(pv, px, py)]
        fl = new_feature_list(nfeat)
        feats = []
        alive = rng.random(nfeat) < rng.choice([0.0, 0.3, 0.9])
        for f, a in zip(fl, alive):
            if a:
                f.x, f.y, f.val = float(np.float32(rng.uniform(0, ncols - 1))), float(np.float32(rng.uniform(0, nrows - 1))), int(rng.integers(0, 900))
            feats.append([f.x, f.y, f.val])
        big += ((ncols + max(mindist - 1, 0)) // max(mindist, 1)) * ((nrows + max(mindist - 1, 0)) // max(mindist, 1)) * 4 > 128 * 1024
        sgf._enforceMinimumDistance(points, fl, ncols, nrows, mindist, min_eig, overwrite)
        enforce_minimum_distance(points, feats, ncols, nrows, mindist, min_eig, overwrite)
        have = np.array([(f.x, f.y, f.val) for f in fl], np.float64)
        assert np.array_equal(have, np.array(feats, np.float64)), \
            "case %d: %dx%d, %d points, %d features, mindist %d, min_eig %g, overwrite %d" % (case, ncols, nrows, npts, nfeat, mindist, min_eig, overwrite)
    assert big >= 5, "no case with the occupancy grid in global memory"


def test_min_distance_walk_refuses_candidates_outside_the_image():
    """klt_min_distance_walk marked accepted candidates in a grid of ncols x nrows cells without looking at their coordinates -- a
    key outside the image was a write outside the grid (LDS or device memory).  The ABI checks every key now (the reference asserts when
    its walk reaches the point, selectGoodFeatures.py:90-91), and refuses a zero key (the kernel's end mark); the context is usable afterwards."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError

    def key(val, x, y):
        return (int(np.float32(val).view(np.uint32)) << 32) | (x << 16) | y

    c = Context(0)
    try:
        fl = np.zeros(6, FEAT_DTYPE)
        fl["x"], fl["y"], fl["val"] = -1, -1, -1
        good = [key(9.0, 10, 10), key(8.0, 50, 40), key(7.0, 99, 79)]
        for mindist, (ncols, nrows) in ((5, (100, 80)), (1, (3000, 2500))):          # the grid in LDS / in device memory
            out, placed = c.min_distance_walk(good, ncols, nrows, mindist, True, fl)
            assert placed == 3 and out["x"][:3].tolist() == [10, 50, 99]
            for bad in (key(6.0, ncols, 5), key(6.0, 5, nrows), key(6.0, 65535, 65535)):
                with pytest.raises(KltBackendError, match="outside the %d x %d image" % (ncols, nrows)):
                    c.min_distance_walk(good + [bad], ncols, nrows, mindist, True, fl)
            with pytest.raises(KltBackendError, match="zero key"):
                c.min_distance_walk(good[:2] + [0, key(6.0, 1, 1)], ncols, nrows, mindist, True, fl)      # the walk's own end mark
            out, placed = c.min_distance_walk(good, ncols, nrows, mindist, True, fl)
            assert placed == 3
    finally:
        c.close()
