"""The reference-shaped Python API on the device (b): frames kept resident and every change to them
noticed (the exact frame cache, its opt-in trusting mode, random call sequences), `from klt import *` scripts through the compat aliases,
ComputeImagePyramids / KLTPyramid handles, selection on frames too small for the pyramid, the optimistic reuse of a resident frame,
sequential mode, every selection path complete when the call returns, KLTTrackSequence of any length.  (Folded by component from the
round-3 / 4 / 5 files in round 6: the tests are unchanged.)"""
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


# a driver written the way scripts written against the reference are: star imports of the reference's module names, time.clock()
STAR_IMPORT_DRIVER = '''
from __future__ import print_function
from klt import *
from PIL import Image
from selectGoodFeatures import *
from writeFeatures import *
from trackFeatures import *
import time

tc = KLT_TrackingContext()
tc.nSkippedPixels = 0
tc.max_residue = 10.0
KLTPrintTrackingContext(tc)
first, second = Image.open("img0.pgm"), Image.open("img1.pgm")
features = KLTSelectGoodFeatures(tc, first, 50)
for k, f in enumerate(features):
    print("Feature #{0}:  ({1},{2}) with value of {3}".format(k, f.x, f.y, f.val))
KLTWriteFeatureListToPPM(features, first, "feat1.ppm")
calls, started = 0, time.clock()
for _ in range(100):
    KLTTrackFeatures(tc, first, second, features)
    KLTTrackFeatures(tc, second, first, features)
    calls += 2
print("seconds per call", (time.clock() - started) / calls)
print("remaining", KLTCountRemainingFeatures(features))
for k, f in enumerate(features):
    print("Feature #{0}:  ({1},{2}) with value of {3}".format(k, f.x, f.y, f.val))
KLTWriteFeatureListToPPM(features, second, "feat2.ppm")
'''


def test_star_import_driver_through_compat(tmp_path, golden_dir):
    """north_star: "example1.py runs unchanged".  A script that imports the reference's top-level module names with `import *` and
    times itself with time.clock() (/root/reference example1.py:10-14, :17-65 -- the call sequence, not the file) runs in a fresh
    process with only PYTHONPATH pointing at this repository and its compat directory; the two PPM files and the list after the
    200-call ping-pong are the reference's own (tests/golden/example1.npz)."""
    pytest.importorskip("PIL.Image")
    import shutil
    for name in ("img0.pgm", "img1.pgm"):
        shutil.copy(os.path.join(golden_dir, name), tmp_path / name)
    script = tmp_path / "driver.py"
    script.write_text(STAR_IMPORT_DRIVER)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, "pyfeaturetrack_amd", "compat")]))
    r = subprocess.run([sys.executable, str(script)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ex = np.load(os.path.join(golden_dir, "example1.npz"))
    for name in ("feat1", "feat2"):
        data = (tmp_path / (name + ".ppm")).read_bytes()
        assert np.array_equal(np.frombuffer(hashlib.sha256(data).digest(), np.uint8), ex[name + "_ppm_sha"]), name
    feats = [l for l in r.stdout.splitlines() if l.startswith("Feature #")]
    assert len(feats) == 100
    for k in range(50):
        x, y, v = ex["pp_after_200_x"][k], ex["pp_after_200_y"][k], int(ex["pp_after_200_val"][k])
        assert feats[50 + k] == "Feature #{0}:  ({1},{2}) with value of {3}".format(k, float(x), float(y), v), k
    assert "remaining %d" % int((ex["pp_after_200_val"] >= 0).sum()) in r.stdout


@pytest.mark.skipif(bool(os.environ.get("KLT_NO_FRAME_CACHE")), reason="the frame cache was switched off through the environment")
def test_python_api_keeps_frames_resident_and_notices_changes():
    """The reference-shaped API with the frame cache (_frames.py): the ping-pong of example1 uploads and builds nothing after its
    first round trip, results are those of a cache-less run, an image edited in place is seen as new, KLTForgetFrames voids the
    cache, and KLT_NO_FRAME_CACHE=1 (a fresh process) gives the same lists."""
    from helpers import make_tc
    from pyfeaturetrack_amd.backend import default_context
    from pyfeaturetrack_amd._frames import cache_of
    sgf, trk = _api_modules()
    try:
        base = synth.synth_base(640, 480, 21)
        f = [synth.synth_frame(640, 480, 21, k, shift=(1.7, -1.1), base=base) for k in range(3)]
        n = 400

        def run(tc, forget):
            out = []
            fl = sgf.KLTSelectGoodFeatures(tc, f[0], n)
            out.append(_records(fl))
            for k in range(6):                                   # ping-pong between two frames, then a third one
                a, b = (f[0], f[1]) if k % 2 == 0 else (f[1], f[0])
                if forget:
                    trk.KLTForgetFrames(tc)
                trk.KLTTrackFeatures(tc, a, b, fl)
                out.append(_records(fl))
            trk.KLTTrackFeatures(tc, f[0], f[2], fl)
            out.append(_records(fl))
            sgf.KLTReplaceLostFeatures(tc, f[2], fl)             # the image frame 2's slot holds: selected on level 0 of its pyramid
            out.append(_records(fl))
            return out

        tc1, tc2 = make_tc(levels=2, ss=4, max_residue=10.0), make_tc(levels=2, ss=4, max_residue=10.0)
        want = run(tc1, forget=True)
        ctx = default_context()
        ctx.timing_enable(1)
        got = run(tc2, forget=False)
        launches = {k["name"]: k["launches"] for k in ctx.timing_read()}
        ctx.timing_enable(0)
        assert got == want
        # select(f0) builds f0's pyramid, track #1 builds f1's, track #7 builds f2's: three level-0 launches in all
        assert launches.get("smooth_grad_l0") == 3, launches
        assert launches.get("track") == 7
        # an in-place edit of the frame is a new frame
        tc3 = make_tc(levels=2, ss=4, max_residue=10.0)
        g0, g1 = f[0].copy(), f[1].copy()
        fl = sgf.KLTSelectGoodFeatures(tc3, g0, n)
        trk.KLTTrackFeatures(tc3, g0, g1, fl)
        first = _records(fl)
        g1[:] = f[2]                                             # same object, other pixels
        fl2 = sgf.KLTSelectGoodFeatures(tc3, g0, n)
        trk.KLTTrackFeatures(tc3, g0, g1, fl2)
        tc4 = make_tc(levels=2, ss=4, max_residue=10.0)
        fl3 = sgf.KLTSelectGoodFeatures(tc4, f[0], n)
        trk.KLTTrackFeatures(tc4, f[0], f[2], fl3)
        assert _records(fl2) == _records(fl3) and _records(fl2) != first
        assert len(cache_of(tc3).held) == 2
        # sequential mode keeps working through the cache (frame 2 becomes frame 1)
        tc5, tc6 = make_tc(levels=2, ss=4), make_tc(levels=2, ss=4)
        tc5.sequentialMode = True
        a = sgf.KLTSelectGoodFeatures(tc5, f[0], n)
        b = sgf.KLTSelectGoodFeatures(tc6, f[0], n)
        for k in (1, 2):
            trk.KLTTrackFeatures(tc5, f[k - 1], f[k], a)
            trk.KLTTrackFeatures(tc6, f[k - 1], f[k], b)
            assert _records(a) == _records(b), k
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def test_python_api_without_the_frame_cache_gives_the_same_example1(tmp_path, golden_dir):
    """KLT_NO_FRAME_CACHE=1: every call uploads and rebuilds what it is given, as the reference does; example1's files and lists
    are the same."""
    env = dict(os.environ, KLT_NO_FRAME_CACHE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "example1.py"), "--out", str(tmp_path), "--iterations", "10"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ex = np.load(os.path.join(golden_dir, "example1.npz"))
    data = (tmp_path / "feat1.ppm").read_bytes()
    assert np.array_equal(np.frombuffer(hashlib.sha256(data).digest(), np.uint8), ex["feat1_ppm_sha"])
    last = [l for l in r.stdout.splitlines() if l.startswith("Feature #49:")][-1]
    x, y, v = ex["pp_after_20_x"][49], ex["pp_after_20_y"][49], int(ex["pp_after_20_val"][49])
    assert last == "Feature #49:  ({0},{1}) with value of {2}".format(float(x), float(y), v)


def test_frame_cache_on_random_call_sequences():
    """Random sequences of KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures over a pool of five frames (some of them
    edited in place between calls), with and without sequential mode: a tracking context that remembers what its slots hold gives
    the same lists, call by call, as one that forgets before every call (= the reference's behaviour of converting and rebuilding
    everything every time)."""
    from helpers import make_tc
    sgf, trk = _api_modules()
    rng = np.random.default_rng(5)
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
            for step in range(10):
                op = rng.choice(["track", "track", "track", "replace", "select", "edit"])
                if op == "edit":                               # same object, new pixels (a block large enough to hold lattice samples)
                    k = int(rng.integers(0, 5))
                    y, x = int(rng.integers(0, 200)), int(rng.integers(0, 300))
                    val = int(rng.integers(0, 255))
                    for fr in (frames_a, frames_b):
                        fr[k][y:y + 60, x:x + 60] = val
                    continue
                nxt = int(rng.integers(0, 5))
                for which, (tc, fr) in enumerate(zip(tcs, (frames_a, frames_b))):
                    if which == 1:
                        trk.KLTForgetFrames(tc)
                    if op == "track":
                        trk.KLTTrackFeatures(tc, fr[cur], fr[nxt], fls[which])
                    elif op == "replace":
                        sgf.KLTReplaceLostFeatures(tc, fr[cur], fls[which])
                    else:
                        fls[which] = sgf.KLTSelectGoodFeatures(tc, fr[nxt], n)
                if op != "replace":
                    cur = nxt
                assert _records(fls[0]) == _records(fls[1]), "trial %d step %d (%s, sequential %s)" % (trial, step, op, seq_mode)
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def _level0_launches(ctx):
    return {k["name"]: k["launches"] for k in ctx.timing_read()}.get("smooth_grad_l0", 0)


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


def test_klt_pyramid_class_on_the_device(cfg1, synth251):
    """KLTPyramid.Compute (pyramid.py:37-77; an API-compatibility class, the tracker builds its pyramids with klt_build_pyramids):
    all levels in one call on the device, equal to the reference's pyramid planes -- level 1 of img0's and img1's image pyramids from
    their level 0 (goldens of the reference itself), and to the levels of the tracker's own build at an odd size with three levels."""
    from helpers import make_tc, synth251_frames
    from pyfeaturetrack_amd.backend import Context
    from pyfeaturetrack_amd.pyramid import KLTPyramid
    for name in ("p0", "p1"):
        lvl0 = cfg1[name + "_img_0"]
        pyr = KLTPyramid(lvl0.shape[1], lvl0.shape[0], 4, 2)
        pyr.Compute(lvl0, 0.9)
        assert np.array_equal(pyr.img[0], lvl0) and np.array_equal(pyr.img[1], cfg1[name + "_img_1"])
        assert pyr.ncols == [320, 80.0] and pyr.nrows == [240, 60.0]
    tc = make_tc(levels=3, ss=2)
    c = Context(0)
    try:
        c.configure(tc)
        c.upload(0, synth251_frames()[0])
        c.build_pyramids(0)
        want = [c.download_level(0, 0, l) for l in range(3)]
    finally:
        c.close()
    pyr = KLTPyramid(251, 187, 2, 3)
    pyr.Compute(want[0], tc.pyramid_sigma_fact)
    for l in range(3):
        assert np.array_equal(pyr.img[l], want[l]), l
    one = KLTPyramid(64, 48, 4, 1)
    one.Compute(np.ones((48, 64), np.float32), 0.9)
    assert len(one.img) == 1


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


# four tracking contexts that differ in everything the device context caches: window, levels, subsampling, frame size; two of them
# use the SAME feature count (the pinned record buffers are cached per length)
_CASES = [
    dict(size=(320, 240), n=120, tc=dict(levels=2, ss=4, window=7, max_residue=10.0)),
    dict(size=(648, 486), n=300, tc=dict(levels=3, ss=2, window=9)),
    dict(size=(500, 380), n=300, tc=dict(levels=2, ss=2, window=5, max_residue=12.0)),
    dict(size=(960, 540), n=700, tc=dict(levels=3, ss=4, window=11)),
]


def _frames_of(k, rounds):
    w, h = _CASES[k]["size"]
    base = synth.synth_base(w, h, 40 + k)
    return [synth.synth_frame(w, h, 40 + k, r, shift=(1.7, -1.1), base=base) for r in range(rounds + 1)]


@default_cache
def test_optimistic_reuse_repeats_the_tracker_when_a_frame_was_edited_in_place():
    """KLTTrackFeatures enqueues the tracker on the strength of size + lattice and compares every byte while the device runs; a frame
    that was edited in place OFF the lattice (frame 1, frame 2, or both) is sent, rebuilt and tracked again inside the same call: the
    lists are those of a tracking context that has never seen the frames."""
    sgf, trk = _api_modules()
    w, h, n = 648, 486, 400
    base = synth.synth_base(w, h, 5)
    f0 = synth.synth_frame(w, h, 5, 0, shift=(2.2, -1.4), base=base)
    f1 = synth.synth_frame(w, h, 5, 1, shift=(2.2, -1.4), base=base)
    tc = make_tc(levels=2, ss=4, max_residue=10.0)
    fl = sgf.KLTSelectGoodFeatures(tc, f0, n)
    trk.KLTTrackFeatures(tc, f0, f1, fl)

    def fresh(a, b):
        t = make_tc(levels=2, ss=4, max_residue=10.0)
        l = sgf.KLTSelectGoodFeatures(t, a, n)
        trk.KLTTrackFeatures(t, a, b, l)
        return _records(l)

    rng = np.random.default_rng(3)
    for edit in ("second", "first", "both", "none", "second"):
        for img in {"second": (f1,), "first": (f0,), "both": (f0, f1), "none": ()}[edit]:
            # a block of pixels off the 32 x 32 lattice (rows / columns that are no multiples of the lattice strides), strong enough to move features
            y, x = int(rng.integers(40, h - 60)) | 1, int(rng.integers(40, w - 60)) | 1
            img[y:y + 9:2, x:x + 9:2] ^= 0x5A
        fl = sgf.KLTSelectGoodFeatures(tc, f0, n)
        trk.KLTTrackFeatures(tc, f0, f1, fl)
        assert _records(fl) == fresh(f0.copy(), f1.copy()), "after editing %s" % edit


def test_sequential_mode_downloads_no_plane_nobody_looks_at(monkeypatch):
    """tc.pyramid_last* are replaced on every sequential-mode KLTTrackFeatures call (trackFeatures.py:401-404).  The handles dropped that
    way must be gone before their slot is overwritten: with a reference cycle inside them they lingered until the cycle collector ran,
    and every call downloaded the nine planes of the call before (4.7 instead of 0.45 ms per 1080p frame through the per-frame API).
    A handle somebody KEEPS still gets its planes before the slot is reused."""
    import gc
    from pyfeaturetrack_amd.backend import Context
    sgf, trk = _api_modules()
    k = 3
    frames = _frames_of(k, 9)
    tc = make_tc(**_CASES[k]["tc"])
    tc.sequentialMode = True
    fl = sgf.KLTSelectGoodFeatures(tc, frames[0], _CASES[k]["n"])
    calls = []
    real = Context.download_level
    monkeypatch.setattr(Context, "download_level", lambda self, *a: (calls.append(a), real(self, *a))[1])
    gc.disable()                                                   # nothing but reference counting may free the dropped handles
    try:
        for r in range(1, 7):
            trk.KLTTrackFeatures(tc, frames[r - 1], frames[r], fl)
            sgf.KLTReplaceLostFeatures(tc, frames[r], fl)
        assert calls == [], "%d planes were downloaded although nobody kept a pyramid handle" % len(calls)
        kept = tc.pyramid_last                                     # ... but a kept handle survives the next two frames with its planes
        trk.KLTTrackFeatures(tc, frames[6], frames[7], fl)
        trk.KLTTrackFeatures(tc, frames[7], frames[8], fl)
        assert len(calls) == tc.nPyramidLevels and kept.img[0].shape == frames[0].shape
    finally:
        gc.enable()
    tc2 = make_tc(**_CASES[k]["tc"])
    want = trk.ComputeImagePyramids(tc2, frames[5], frames[6])[3]
    assert np.array_equal(kept.img[0], want.img[0]) and np.array_equal(kept.img[tc.nPyramidLevels - 1], want.img[tc.nPyramidLevels - 1])


def test_api_selection_is_complete_when_it_returns_on_every_selection_path():
    """The API's selection lists are pinned host memory the kernels write in place (klt_featbuf_map_host); the call must not return before
    the LAST kernel has written them on ANY path of the selection: the parallel passes (klt_select_finish waits), the sorted serial walk
    (KLT_OPT_SELECT_PARALLEL_NMS = 0, or an exclusion square too large for the passes' tile: completes inside klt_select_begin_async
    without a host wait), and a frame without a single candidate.  Found by tests/fuzz/fuzz_seeds_r05.sh (1 of 15 000 sequence trials
    differed, not reproducibly).  Every call is compared with the synchronous ABI call on a context of its own; stale records of the
    call before (another frame, the same list length) sit in the mapped array each time."""
    from pyfeaturetrack_amd.backend import Context, context_of
    from pyfeaturetrack_amd.params import params_from_tc
    sgf, trk = _api_modules()
    ref = Context(0)
    try:
        cases = [dict(size=(640, 480), tc=dict(levels=2, ss=4, mindist=10), serial=False),
                 dict(size=(640, 480), tc=dict(levels=2, ss=4, mindist=10), serial=True),
                 dict(size=(900, 700), tc=dict(levels=2, ss=2, mindist=130), serial=False),         # the passes' tile would not fit: serial walk
                 dict(size=(168, 553), tc=dict(levels=4, ss=2, window=9, mindist=10), serial=False)]   # border 84: no candidate column
        for case in cases:
            w, h = case["size"]
            tc = make_tc(**case["tc"])
            frames = [synth.synth_frame(w, h, 77, k, shift=(2.0, 1.0)) for k in range(4)]
            ctx = None
            for rep in range(12):
                f = frames[rep % 4]
                if ctx is not None:
                    ctx.set_option(8, 0 if case["serial"] else 1)
                fl = sgf.KLTSelectGoodFeatures(tc, f, 150)
                ctx = context_of(tc)
                ref.configure(tc)
                ref.set_option(8, 0 if case["serial"] else 1)
                ref.upload(0, f)
                ref.build_pyramids(0)
                want, _ = ref.select(0, 150, use_pyramid=True)
                have = np.array([(a.x, a.y, a.val) for a in fl], np.float64)
                assert (np.array_equal(have[:, 2], want["val"].astype(np.float64)) and np.array_equal(have[:, 0], want["x"].astype(np.float64))
                        and np.array_equal(have[:, 1], want["y"].astype(np.float64))), (case, rep)
                # ... and a replacement on the next frame through the same mapped array
                g = frames[(rep + 1) % 4]
                trk.KLTTrackFeatures(tc, f, g, fl)
                sgf.KLTReplaceLostFeatures(tc, g, fl)
                ref.upload(1, g)
                ref.build_pyramids(1)
                out, _ = ref.track(0, 1, want)
                rep_want, _ = ref.select(1, 150, mode=2, fl=out, use_pyramid=True)
                have = np.array([(a.x, a.y, a.val) for a in fl], np.float64)
                assert np.array_equal(have[:, 2], rep_want["val"].astype(np.float64)) and np.array_equal(have[:, 0], rep_want["x"].astype(np.float64)), (case, rep, "replacement")
            ctx.set_option(8, 1)
    finally:
        ref.close()


def test_python_api_over_random_call_sequences_vs_oracle():
    """tests/fuzz/fuzz_parity.py --api: KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures on one tracking context in random
    order over four frames that are edited in place between calls, a third of the draws in sequential mode -- every list equals the
    ORACLE's after every call (the exact frame cache with its optimistic device work, lists mapped into pinned memory, recycled feature
    objects, scores prepared ahead: none of it may show).  8400 draws ran when the mode was written (profiles/history/r05_fuzz_seeds.txt); 25
    stay in the suite."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(99)
    for k in range(25):
        t = fz.draw(rng, 250000, 500, 700)
        bad = fz.run_api_trial(t)
        assert bad is None, "draw %d: %s differs from the oracle: %r" % (k, bad, t)


@pytest.mark.timeout(300)
def test_track_sequence_of_any_length_and_a_slow_frame_source():
    """KLTTrackSequence keeps frames on their way two steps ahead of the tracker: sequences of 1 .. 7 frames (the frame source ends
    during the start-up sends -- a ONE-frame sequence used to ask the helper thread for a frame after it had said "no more", and waited
    for ever), with frames that arrive late (a generator that sleeps) and early, with and without the helper thread, equal the per-frame
    API loop row by row; so does every further call on the same tracking context and the per-frame call that continues the sequence."""
    import time
    from test_gpu_parity import _host_api_sequence
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n = 360, 280, 250
    base = synth.synth_base(w, h, 23)
    frames = [synth.synth_frame(w, h, 23, k, shift=(2.1, -1.7), base=base) for k in range(7)]
    frames[3] = frames[3].copy()
    frames[3][60:150, 100:240] = 128                      # features there are lost and replaced elsewhere

    def make():
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        tc.max_residue = 10.0
        return tc

    def slow(seq, every):
        for k, f in enumerate(seq):
            if every and k % every == every - 1:
                time.sleep(0.02)                          # the look comes long before this frame is staged
            yield f

    sgf.KLT_verbose = 0
    try:
        tc = make()
        for nf in (1, 2, 3, 4, 5, 7):
            want = _host_api_sequence(make(), frames[:nf], n, True)
            for ingest, every in ((True, 0), (True, 2), (True, 1), (False, 0)):
                got = KLTTrackSequence(tc, slow(frames[:nf], every), n, replace_lost=True, async_ingest=ingest)
                assert got.nFrames == nf, (nf, ingest, every)
                assert np.array_equal(got.val, want.val) and np.array_equal(got.x, want.x) and np.array_equal(got.y, want.y), (nf, ingest, every)
        if True:
            from pyfeaturetrack_amd import storeFeatures as sf
            from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
            nxt = synth.synth_frame(w, h, 23, 7, shift=(2.1, -1.7), base=base)
            fl, fl2 = sf.KLTCreateFeatureList(n), sf.KLTCreateFeatureList(n)
            sf.KLTExtractFeatureList(fl, got, 6)
            sf.KLTExtractFeatureList(fl2, got, 6)
            KLTTrackFeatures(tc, frames[6], nxt, fl)
            other = make()
            other.sequentialMode = False
            KLTTrackFeatures(other, frames[6], nxt, fl2)
            assert [(f.x, f.y, f.val) for f in fl] == [(f.x, f.y, f.val) for f in fl2]
    finally:
        sgf.KLT_verbose = 1


def test_track_sequence_stages_frames_of_4_mb_and_more_on_two_helper_threads(monkeypatch):
    """KLTTrackSequence on 2400 x 1800 frames (4.3 MB: `trackSequence.STAGER_THREADS_FROM_BYTES`): two helper threads copy whole frames side
    by side and deliver them in order; the table is the one ONE helper gives, the one the synchronous ingest gives, and -- frame by frame --
    the per-frame API's; a generator of uneven speed and a Pillow image in the clip change nothing (round 6)."""
    import time
    from PIL import Image
    from pyfeaturetrack_amd import storeFeatures as sf, trackSequence
    sgf, trk = _api_modules()
    w, h, n, nf = 2400, 1800, 1500, 11
    assert w * h >= trackSequence.STAGER_THREADS_FROM_BYTES
    base = synth.synth_base(w, h, 5)
    frames = [synth.synth_frame(w, h, 5, k, shift=(2.3, -1.4), base=base) for k in range(nf)]

    def tc_of():
        tc = make_tc(levels=3, ss=4, max_residue=10.0)
        tc.sequentialMode = True
        return tc

    seen = []
    real = trackSequence._FrameStager

    class Watched(real):
        def __init__(self, frames_, buffers, shape, workers=1):
            seen.append((workers, len(buffers)))
            real.__init__(self, frames_, buffers, shape, workers=workers)
    monkeypatch.setattr(trackSequence, "_FrameStager", Watched)

    def uneven():
        for k, f in enumerate(frames):
            time.sleep(0.004 if k % 3 == 1 else 0.0)
            yield Image.fromarray(f) if k == 4 else f

    two = trackSequence.KLTTrackSequence(tc_of(), uneven(), n)
    assert seen[-1][0] == 2 and seen[-1][1] >= 6
    monkeypatch.setattr(trackSequence, "STAGER_WORKERS", 1)
    one = trackSequence.KLTTrackSequence(tc_of(), iter(frames), n)
    assert seen[-1][0] == 1
    sync = trackSequence.KLTTrackSequence(tc_of(), iter(frames), n, async_ingest=False)
    assert np.array_equal(two.rec, one.rec) and np.array_equal(two.rec, sync.rec)
    tc = tc_of()
    want = sf.KLTCreateFeatureTable(nf, n)
    fl = sgf.KLTSelectGoodFeatures(tc, frames[0], n)
    sf.KLTStoreFeatureList(fl, want, 0)
    for k in range(1, nf):
        trk.KLTTrackFeatures(tc, frames[k - 1], frames[k], fl)
        sgf.KLTReplaceLostFeatures(tc, frames[k], fl)
        sf.KLTStoreFeatureList(fl, want, k)
    assert np.array_equal(two.val, want.val) and np.array_equal(two.x, want.x) and np.array_equal(two.y, want.y)
