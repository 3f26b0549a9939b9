"""Round-4 GPU checks: the exact frame cache of the reference-shaped API, ComputeImagePyramids and its device-resident pyramid
handles, the reference's literal native boundary (compat goodFeaturesUtils / trackFeaturesUtils) and INTEGRATION.md's stub."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from helpers import make_tc
from pyfeaturetrack_amd import synth

pytestmark = pytest.mark.gpu


def _api_modules():
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import trackFeatures as trk
    sgf.KLT_verbose = trk.KLT_verbose = 0
    return sgf, trk


def _records(fl):
    return [(f.x, f.y, f.val) for f in fl]


def _level0_launches(ctx):
    return {k["name"]: k["launches"] for k in ctx.timing_read()}.get("smooth_grad_l0", 0)


# These tests are about the frame cache's DEFAULT behaviour (exact reuse); the environment switches that change the default for a whole
# process (KLT_NO_FRAME_CACHE, KLT_TRUST_FRAME_IDENTITY) make their expectations meaningless, not wrong.
default_cache = pytest.mark.skipif(bool(os.environ.get("KLT_NO_FRAME_CACHE") or os.environ.get("KLT_TRUST_FRAME_IDENTITY")),
                                   reason="the frame cache's default was changed through the environment")


# ------------------------------------------------------------------------------------------------ frame cache: exact by default
@default_cache
@pytest.mark.parametrize("kind", ["numpy", "pil"])
def test_in_place_edits_of_any_size_are_seen(kind):
    """the reference converts and rebuilds both images on every call (trackFeatures.py:163-176).  The frame cache
    may skip that only for an image with exactly the pixels a slot holds: ONE pixel rewritten in place off the old 32 x 32 lattice,
    and a 32 x 59 block between lattice samples, both give the lists of a run that forgets everything before every call -- for
    numpy frames and Pillow images.  (1080p: lattice rows are multiples of 33, lattice columns multiples of 60.)"""
    from pyfeaturetrack_amd.backend import default_context
    sgf, trk = _api_modules()
    if kind == "pil":
        Image = pytest.importorskip("PIL.Image")
    try:
        W, H, n = 1920, 1080, 1500
        base = synth.synth_base(W, H, 3)
        f0, f1 = (synth.synth_frame(W, H, 3, k, shift=(2.3, -1.4), base=base) for k in range(2))

        def wrap(a):
            return Image.fromarray(a.copy()) if kind == "pil" else a.copy()

        def edit_pixel(img, x, y):
            if kind == "pil":
                img.putpixel((x, y), 255 - img.getpixel((x, y)))
            else:
                img[y, x] = 255 - img[y, x]

        def edit_block(img, x, y, w, h):
            if kind == "pil":
                img.paste(7, (x, y, x + w, y + h))
            else:
                img[y:y + h, x:x + w] = 7

        def run(forget):
            tc = make_tc(levels=3, ss=4, max_residue=10.0)
            a, b = wrap(f0), wrap(f1)
            out = []
            fl = sgf.KLTSelectGoodFeatures(tc, a, n)
            trk.KLTTrackFeatures(tc, a, b, fl)
            out.append(_records(fl))
            # a feature that survived: the edits go under its 7 x 7 window in frame 2, at off-lattice coordinates
            live = [f for f in fl if f.val >= 0 and int(f.x) % 60 not in (0, 59, 58, 57) and int(f.y) % 33 not in (0, 32, 31, 30)]
            cx, cy = int(live[0].x), int(live[0].y)
            assert cx % 60 != 0 and cy % 33 != 0
            steps = [lambda: edit_pixel(b, cx, cy),
                     lambda: edit_block(b, 61 + 60 * (cx // 60 % 20), 34 + 33 * (cy // 33 % 20), 59, 32),
                     lambda: edit_pixel(a, cx + 1, cy)]
            for step in steps:
                step()
                if forget:
                    trk.KLTForgetFrames(tc)
                fl = sgf.KLTSelectGoodFeatures(tc, a, n)
                if forget:
                    trk.KLTForgetFrames(tc)
                trk.KLTTrackFeatures(tc, a, b, fl)
                out.append(_records(fl))
            return out

        want = run(forget=True)
        ctx = default_context()
        ctx.timing_enable(1)
        got = run(forget=False)
        builds = _level0_launches(ctx)
        ctx.timing_enable(0)
        assert got == want
        assert got[1] != got[0], "the one-pixel edit under a feature window changed nothing: the test does not probe the cache"
        # frame 1 (for the selection), frame 2 (for the tracker), then one rebuild per edit -- only the edited image each time
        assert builds == 5, builds
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


@default_cache
def test_the_trusting_mode_is_opt_in_and_blind_between_lattice_samples():
    """tc.trustFrameIdentity = True is the documented shortcut: object identity + the 1024-pixel lattice.  It does
    NOT see an off-lattice edit -- which is why it is not the default -- and KLTForgetFrames makes it look again."""
    from pyfeaturetrack_amd.backend import default_context
    sgf, trk = _api_modules()
    try:
        W, H, n = 1920, 1080, 600
        base = synth.synth_base(W, H, 5)
        a, b = (synth.synth_frame(W, H, 5, k, shift=(1.2, 0.9), base=base) for k in range(2))
        tc = make_tc(levels=3, ss=4)
        tc.trustFrameIdentity = True
        fl = sgf.KLTSelectGoodFeatures(tc, a, n)
        trk.KLTTrackFeatures(tc, a, b, fl)
        ctx = default_context()
        ctx.timing_enable(1)
        b[100:132, 61:120] = 9
        fl2 = sgf.KLTSelectGoodFeatures(tc, a, n)
        trk.KLTTrackFeatures(tc, a, b, fl2)
        assert _level0_launches(ctx) == 0 and _records(fl2) == _records(fl)          # the stale pyramid: the deviation
        trk.KLTForgetFrames(tc)
        trk.KLTTrackFeatures(tc, a, b, sgf.KLTSelectGoodFeatures(tc, a, n))
        assert _level0_launches(ctx) == 2                                            # both frames again
        ctx.timing_enable(0)
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


@default_cache
def test_frame_cache_on_random_call_sequences_with_arbitrary_edits():
    """Random sequences of KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures / ComputeImagePyramids over a pool of five
    frames, edited in place between calls by rectangles of ANY size and position (down to one pixel), with and without sequential
    mode: a tracking context that remembers what its slots hold gives the same lists, call by call, as one that forgets before every
    call (= the reference's behaviour of converting and rebuilding everything every time)."""
    sgf, trk = _api_modules()
    rng = np.random.default_rng(17)
    try:
        base = synth.synth_base(400, 300, 8)
        pool = [synth.synth_frame(400, 300, 8, k, shift=(1.3, 0.9), base=base) for k in range(5)]
        for trial in range(40):
            seq_mode = bool(trial & 1)
            frames_a = [f.copy() for f in pool]
            frames_b = [f.copy() for f in pool]
            tcs = []
            for _ in range(2):
                tc = make_tc(levels=2, ss=2, max_residue=12.0)
                tc.sequentialMode = seq_mode
                tcs.append(tc)
            n = int(rng.integers(40, 200))
            i0 = int(rng.integers(0, 5))
            fls = [sgf.KLTSelectGoodFeatures(tcs[0], frames_a[i0], n)]
            trk.KLTForgetFrames(tcs[1])
            fls.append(sgf.KLTSelectGoodFeatures(tcs[1], frames_b[i0], n))
            assert _records(fls[0]) == _records(fls[1])
            cur = i0
            for step in range(12):
                op = rng.choice(["track", "track", "track", "replace", "select", "edit", "edit", "pyramids"])
                if op == "edit":                               # same object, new pixels: any rectangle, often tiny
                    k = int(rng.integers(0, 5))
                    h, w = (1, 1) if rng.random() < 0.4 else (int(rng.integers(1, 40)), int(rng.integers(1, 70)))
                    y, x = int(rng.integers(0, 300 - h)), int(rng.integers(0, 400 - w))
                    val = int(rng.integers(0, 255))
                    for fr in (frames_a, frames_b):
                        fr[k][y:y + h, x:x + w] = val
                    continue
                nxt = int(rng.integers(0, 5))
                planes = []
                for which, (tc, fr) in enumerate(zip(tcs, (frames_a, frames_b))):
                    if which == 1:
                        trk.KLTForgetFrames(tc)
                    if op == "track":
                        trk.KLTTrackFeatures(tc, fr[cur], fr[nxt], fls[which])
                    elif op == "replace":
                        sgf.KLTReplaceLostFeatures(tc, fr[cur], fls[which])
                    elif op == "pyramids":
                        pyr = trk.ComputeImagePyramids(tc, fr[cur], fr[nxt])
                        planes.append([p.img[tc.nPyramidLevels - 1] for p in pyr])
                    else:
                        fls[which] = sgf.KLTSelectGoodFeatures(tc, fr[nxt], n)
                if planes:
                    assert all(np.array_equal(p, q) for p, q in zip(*planes)), "trial %d step %d pyramids" % (trial, step)
                if op in ("track", "select"):
                    cur = nxt
                assert _records(fls[0]) == _records(fls[1]), "trial %d step %d (%s, sequential %s)" % (trial, step, op, seq_mode)
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


# ------------------------------------------------------------------------------------------------ ComputeImagePyramids
def test_compute_image_pyramids_gives_the_reference_planes(cfg1, img0, img1):
    """trackFeatures.py:146-196 as a callable name (`from trackFeatures import *` exposes it): the six pyramids of img0 / img1 with the
    reference's KLTPyramid attributes; every level of every pyramid equals the planes the reference produced (tests/golden/cfg1.npz,
    written by gen_golden.py from the reference's own ComputeImagePyramids)."""
    sgf, trk = _api_modules()
    try:
        tc = make_tc()
        pyr = trk.ComputeImagePyramids(tc, img0, img1)
        assert len(pyr) == 6
        for p in pyr:
            assert p.subsampling == 4 and p.nLevels == 2 and p.ncols == [320, 80.0] and p.nrows == [240, 60.0] and len(p.img) == 2
        for which, frame in ((0, "p0"), (3, "p1")):
            for k, name in enumerate(("img", "gx", "gy")):
                for lvl in range(2):
                    got = pyr[which + k].img[lvl]
                    assert got.dtype == np.float32 and np.array_equal(got, cfg1["%s_%s_%d" % (frame, name, lvl)]), (frame, name, lvl)
        # star-import name, like the reference's module
        ns = {}
        exec("from pyfeaturetrack_amd.trackFeatures import *", ns)
        assert ns["ComputeImagePyramids"] is trk.ComputeImagePyramids
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def test_pyramid_handles_outlive_the_slot_and_follow_sequential_mode(cfg1, img0, img1):
    """The handles download on access -- and a handle somebody kept is filled before its slot is overwritten (a new frame, new
    parameters, a sequence call), so it stays valid like the reference's pyramid objects.  In sequential mode the first three pyramids
    are tc.pyramid_last* and img1 is ignored (trackFeatures.py:152-161)."""
    sgf, trk = _api_modules()
    try:
        tc = make_tc()
        kept = trk.ComputeImagePyramids(tc, img0, img1)             # nothing downloaded yet
        other = np.ascontiguousarray(img0[::-1])
        fl = sgf.KLTSelectGoodFeatures(tc, other, 30)                # overwrites slot 1 (frame img0)
        trk.KLTTrackFeatures(tc, other, np.ascontiguousarray(img1[::-1]), fl)   # ... and slot 2
        assert np.array_equal(kept[0].img[1], cfg1["p0_img_1"]) and np.array_equal(kept[5].img[0], cfg1["p1_gy_0"])
        kept2 = trk.ComputeImagePyramids(tc, img0, img1)
        tc.nPyramidLevels = 3
        tc.subsampling = 2
        tc.KLTUpdateTCBorder()
        p3 = trk.ComputeImagePyramids(tc, img0, img1)                # new geometry: every pyramid of the context is rebuilt
        assert p3[0].nLevels == 3 and p3[0].img[2].shape == (60, 80)
        assert np.array_equal(kept2[1].img[1], cfg1["p0_gx_1"]) and kept2[1].nLevels == 2

        seq = make_tc()
        seq.sequentialMode = True
        fl = sgf.KLTSelectGoodFeatures(seq, img0, 50)
        trk.KLTTrackFeatures(seq, img0, img1, fl)                    # pyramid_last := pyramids of img1
        assert seq.pyramid_last.ncols[0] == 320 and np.array_equal(seq.pyramid_last_gradx.img[1], cfg1["p1_gx_1"])
        junk = np.zeros_like(img0)
        six = trk.ComputeImagePyramids(seq, junk, img0)              # img1 argument ignored: pyramid 1 is the kept one
        assert six[0] is seq.pyramid_last and six[2] is seq.pyramid_last_grady
        assert np.array_equal(six[0].img[0], cfg1["p1_img_0"]) and np.array_equal(six[3].img[1], cfg1["p0_img_1"])
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


# ------------------------------------------------------------------------------------------------ the reference's literal native boundary
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


# ------------------------------------------------------------------------------------------------ INTEGRATION.md section B
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


# ------------------------------------------------------------------------------------------------ ADVICE r3 (low)
def test_selection_on_a_frame_too_small_for_the_pyramid(img0):
    """KLTSelectGoodFeatures never builds a pyramid in the reference (selectGoodFeatures.py:183-197), so it succeeds on a
    frame the tracking context's pyramid does not fit -- here 8 levels of subsampling 4 on 320x240 (level 4 would be 1x0 pixels).  The
    level-0 shortcut must not turn that into an error; the list is the one a fitting pyramid geometry gives with the same border."""
    sgf, trk = _api_modules()
    try:
        small = make_tc()
        small.nPyramidLevels, small.subsampling = 8, 4
        small.borderx = small.bordery = 30.0
        fits = make_tc()                                       # 2 levels of 4: border 30.0 as well
        assert (fits.borderx, fits.bordery) == (30.0, 30.0)
        a = sgf.KLTSelectGoodFeatures(small, img0, 80)
        b = sgf.KLTSelectGoodFeatures(fits, img0, 80)
        assert _records(a) == _records(b) and sum(f.val >= 0 for f in a) == 80
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def test_dispatch_timestamps_never_drop_a_launch_silently():
    """klt_timing_enable(ctx, 2) times single-launch families by their dispatch's own timestamps, which only launches that
    go through klt_launch carry.  On a frame whose width is not a multiple of 4 the summed-area passes take the barrier-coupled
    fallback kernels: every scope is either measured or counted in "<family>!unstamped" -- never lost."""
    from helpers import synth251_frames
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        for frame, tc in ((synth251_frames()[0], make_tc(levels=2, ss=2)), (synth.synth_pair(640, 480, 2)[0], make_tc(levels=2, ss=4))):
            c.configure(tc)
            c.upload(0, frame)
            c.build_pyramids(0)
            for mode in (1, 2):
                c.timing_enable(mode)
                c.select(0, 40, use_pyramid=True)
                t = {k["name"]: k["launches"] for k in c.timing_read()}
                c.timing_enable(0)
                for fam in ("sat_rows", "sat_cols"):
                    assert t.get(fam, 0) + t.get(fam + "!unstamped", 0) == 1, (frame.shape, mode, t)
                assert mode == 2 or not any(k.endswith("!unstamped") for k in t)
    finally:
        c.close()


# ------------------------------------------------------------------------------------------------ frames already in device memory
def test_a_slot_adopts_a_frame_that_is_already_on_the_device(img0, img1, cfg1):
    """klt_slot_adopt_u8 (zero-copy ingest): a clip kept in device memory (klt_device_alloc / klt_device_write) is read in
    place -- the slot that adopts a frame gives the reference's pyramid planes, selection and tracking, exactly as the slot that was
    uploaded to; an upload into the slot ends the adoption, freeing the clip leaves no slot pointing into it, and the argument checks hold."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(max_residue=10.0))
        n = img0.size
        clip = c.device_alloc(2 * n)
        c.device_write(clip, img0)
        c.device_write(clip + n, img1)
        c.adopt_u8(0, clip, 320, 240)
        c.adopt_u8(1, clip + n, 320, 240)
        assert c.frame_resident(0) and not c.pyramids_valid(0)
        c.build_pyramids_batch([0, 1], sync=True)
        for slot, name in ((0, "p0"), (1, "p1")):
            for l in range(2):
                for pi, w in enumerate(("img", "gx", "gy")):
                    assert np.array_equal(c.download_level(slot, pi, l), cfg1["%s_%s_%d" % (name, w, l)]), (name, w, l)
        fl, placed = c.select(0, 100)                              # from the raw (adopted) frame, not the pyramid
        assert placed == 100 and np.array_equal(fl["x"].astype(np.float64), cfg1["sel100_x"]) and np.array_equal(fl["val"].astype(np.int64), cfg1["sel100_val"])
        out, _ = c.track(0, 1, fl)
        assert np.array_equal(out["val"].astype(np.int64), cfg1["trk100_r10_val"])
        ok = out["val"] >= 0
        assert np.array_equal(out["x"][ok].astype(np.float64), cfg1["trk100_r10_x"][ok])
        c.upload(0, img1)                                          # an upload ends the adoption: the clip's first frame is untouched
        c.adopt_u8(2, clip, 320, 240)
        c.build_pyramids_batch([0, 2], sync=True)
        assert np.array_equal(c.download_level(0, 0, 1), cfg1["p1_img_1"]) and np.array_equal(c.download_level(2, 0, 1), cfg1["p0_img_1"])
        with pytest.raises(KltBackendError):
            c.adopt_u8(3, clip, 320, 240 * 100000)
        with pytest.raises(KltBackendError):
            c._check(c._lib.klt_slot_adopt_u8(c._h, 3, img0.ctypes.data, 320, 240, 320))     # host memory is not adoptable
        with pytest.raises(KltBackendError):
            c._check(c._lib.klt_slot_adopt_u8(c._h, 3, clip, 300, 240, 320))                  # rows must be contiguous
        c.device_free(clip)
        assert not c.frame_resident(2) and not c.frame_resident(1)
        with pytest.raises(KltBackendError):
            c.device_free(clip)
    finally:
        c.close()


def test_the_resident_clip_example_runs():
    """examples/resident_clip.py: the cfg-5 loop on frames read in place from device memory, at a small size."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "resident_clip.py"), "--frames", "12", "--size", "640x480", "--features", "300"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "300 of 300 alive at the end" in r.stdout, r.stdout


# ------------------------------------------------------------------------------------------------ degenerate inputs
def _degenerate_pairs(w, h):
    rng = np.random.default_rng(123)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    checker = (((xx // 8 + yy // 8) & 1) * 255).astype(np.uint8)
    spike = np.zeros((h, w), np.uint8)
    spike[h // 2, w // 2] = 255
    ramp = (xx * 255 // (w - 1)).astype(np.uint8)
    flat = np.full((h, w), 128, np.uint8)
    return {
        "constant -> constant": (flat, flat.copy()),
        "black -> white": (np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8)),
        "white noise, uncorrelated": (noise, np.roll(noise[::-1], 7, axis=1).copy()),
        "white noise, shifted by (2, -1)": (noise, np.roll(noise, (-1, 2), axis=(0, 1))),
        "saturated checkerboard, shifted by half a period": (checker, np.roll(checker, 4, axis=1)),
        "one bright pixel that vanishes": (spike, np.zeros((h, w), np.uint8)),
        "horizontal ramp (no corner anywhere)": (ramp, np.roll(ramp, 3, axis=1)),
        "texture -> constant": (noise, flat),
    }


@pytest.mark.parametrize("levels,ss,window", [(2, 4, 7), (3, 2, 5), (1, 2, 15)])
def test_degenerate_frames_match_the_oracle(levels, ss, window):
    """Inputs a synthetic texture never produces: constant frames (every determinant 0: KLT_SMALL_DET, no candidate above the
    eigenvalue floor), saturated noise and checkerboards (ties, equal eigenvalues, aliasing), a single bright pixel, a ramp, and pairs
    whose second frame has nothing in common with the first -- selection, tracking (with and without the residue test) and replacement
    give the oracle's records, status codes included."""
    from oracle import klt_oracle as ko
    from helpers import params_from_tc
    from pyfeaturetrack_amd.backend import Context, REPLACING_SOME
    w, h, n = 200, 152, 120
    c = Context(0)
    try:
        for mr in (None, 6.0):
            tc = make_tc(levels=levels, ss=ss, window=window, max_residue=mr, mindist=6)
            p = params_from_tc(tc)
            c.configure(tc)
            for name, (f0, f1) in _degenerate_pairs(w, h).items():
                c.upload(0, f0)
                c.upload(1, f1)
                c.build_pyramids_batch([0, 1], sync=True)
                fl, placed = c.select(0, n)
                ofl = ko.select_good_features(p, f0.astype(np.float32), n)
                for k in ("val", "x", "y"):
                    assert np.array_equal(fl[k], ofl[k]), (name, "selection", k, mr)
                out, _ = c.track(0, 1, fl)
                ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
                for k in ("val", "x", "y"):
                    assert np.array_equal(out[k], ofl[k]), (name, "tracking", k, mr, np.unique(ofl["val"]))
                rep, _ = c.select(1, n, mode=REPLACING_SOME, fl=out)
                orep = ko.select_good_features(p, f1.astype(np.float32), n, mode=2, fl=ofl)
                for k in ("val", "x", "y"):
                    assert np.array_equal(rep[k], orep[k]), (name, "replacement", k, mr)
    finally:
        c.close()


def test_records_come_back_without_draining_the_pipeline(img0, img1, cfg1):
    """klt_featbuf_download_async / klt_download_wait: the copy is enqueued in stream order behind the tracker that wrote the records and
    lands in pinned host memory; work enqueued behind it does not disturb it; a pageable destination is refused."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(max_residue=10.0))
        c.upload(0, img0)
        c.upload(1, img1)
        c.build_pyramids_batch([0, 1])
        fl, _ = c.select(0, 100)
        c.featbuf_upload(5, fl)
        out = c.pinned_array((100,), FEAT_DTYPE)
        out["val"] = 77
        c.track_async(0, 1, 5, 6, 100)
        c.featbuf_download_async(6, out)
        c.track_async(1, 0, 6, 7, 100)                              # more work behind the copy: reads buffer 6, must not change what was copied
        c.download_wait()
        assert np.array_equal(out["val"].astype(np.int64), cfg1["trk100_r10_val"])
        ok = out["val"] >= 0
        assert np.array_equal(out["x"][ok].astype(np.float64), cfg1["trk100_r10_x"][ok])
        c.download_wait()                                           # nothing pending: returns at once
        with pytest.raises(KltBackendError):
            c.featbuf_download_async(6, np.empty(100, FEAT_DTYPE))
    finally:
        c.close()


def test_one_context_per_host_thread():
    """include/klt_gpu.h: "One context per host thread / device; no shared mutable globals" (ctypes releases the GIL during every call).
    Six threads, a context each, run upload + pyramids + selection + tracking + replacement on their own frames at the same time, eight
    rounds each with a different frame size per thread (so buffers are grown, LDS attributes set and kernels loaded concurrently); every
    round gives exactly what the same calls give on one thread."""
    import threading
    from pyfeaturetrack_amd.backend import Context
    sizes = [(320, 240, 150), (648, 486, 400), (500, 380, 300), (1280, 720, 1500), (402, 302, 200), (960, 540, 900)]
    ROUNDS = 8
    frames = [[synth.synth_pair(w, h, 100 * k + r) for r in range(ROUNDS)] for k, (w, h, _) in enumerate(sizes)]

    def work(k, out, rounds):
        n = sizes[k][2]
        try:
            c = Context(0)
            try:
                c.configure(make_tc(max_residue=10.0, levels=3 if k % 2 else 2, ss=2 if k % 2 else 4))
                for r in range(rounds):
                    f0, f1 = frames[k][r]
                    c.upload(0, f0)
                    c.upload(1, f1)
                    c.build_pyramids_batch([0, 1], sync=False)
                    fl, placed = c.select(0, n)
                    trk, _ = c.track(0, 1, fl)
                    rep, _ = c.select(1, n, mode=2, fl=trk)
                    out.append((placed, fl.tobytes(), trk.tobytes(), rep.tobytes()))
            finally:
                c.close()
        except BaseException as e:          # noqa: BLE001  (re-raised by the main thread)
            out.append(e)

    want = [[] for _ in sizes]
    for k in range(len(sizes)):
        work(k, want[k], 3)
    got = [[] for _ in sizes]
    threads = [threading.Thread(target=work, args=(k, got[k], ROUNDS)) for k in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    for k in range(len(sizes)):
        for e in got[k] + want[k]:
            if isinstance(e, BaseException):
                raise e
        assert len(got[k]) == ROUNDS and got[k][:3] == want[k], "thread %d" % k
        assert any(rec[0] > 0 for rec in got[k])
