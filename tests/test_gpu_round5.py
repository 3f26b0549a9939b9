"""Round-5 GPU checks: the reference-shaped Python API under threads (one device context per host thread, a lock per context), the
calls' optimistic frame reuse (device work enqueued before every byte of a frame has been compared, repeated when the comparison
fails) and the feature objects (KLT_Feature as a (store, row) pair)."""
import threading

import numpy as np
import pytest

from helpers import make_tc
from pyfeaturetrack_amd import synth

pytestmark = pytest.mark.gpu


import os

# tests of DEFAULT behaviours that an environment switch changes for the whole process are meaningless under that switch, not wrong
default_cache = pytest.mark.skipif(bool(os.environ.get("KLT_NO_FRAME_CACHE") or os.environ.get("KLT_TRUST_FRAME_IDENTITY")),
                                   reason="the frame cache's default was changed through the environment")
default_lists = pytest.mark.skipif(bool(os.environ.get("KLT_NO_FEATURE_RECYCLING") == "1" or os.environ.get("KLT_LAZY_FEATURE_LISTS") == "1"),
                                   reason="feature lists are lazy / not recycled through the environment")


def _api_modules():
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import trackFeatures as trk
    sgf.KLT_verbose = trk.KLT_verbose = 0
    return sgf, trk


def _records(fl):
    return [(f.x, f.y, f.val) for f in fl]


# four tracking contexts that differ in everything the device context caches: window, levels, subsampling, frame size; two of them
# use the SAME feature count (the pinned record buffers are cached per length)
_CASES = [
    dict(size=(320, 240), n=120, tc=dict(levels=2, ss=4, window=7, max_residue=10.0)),
    dict(size=(648, 486), n=300, tc=dict(levels=3, ss=2, window=9)),
    dict(size=(500, 380), n=300, tc=dict(levels=2, ss=2, window=5, max_residue=12.0)),
    dict(size=(960, 540), n=700, tc=dict(levels=3, ss=4, window=11)),
]
_ROUNDS = 50


def _frames_of(k, rounds):
    w, h = _CASES[k]["size"]
    base = synth.synth_base(w, h, 40 + k)
    return [synth.synth_frame(w, h, 40 + k, r, shift=(1.7, -1.1), base=base) for r in range(rounds + 1)]


def _api_rounds(k, frames, rounds, tc=None, out=None):
    """select on frame r, track r -> r+1, replace the lost ones on r+1: the three public calls, `rounds` times"""
    sgf, trk = _api_modules()
    tc = tc or make_tc(**_CASES[k]["tc"])
    out = [] if out is None else out
    for r in range(rounds):
        fl = sgf.KLTSelectGoodFeatures(tc, frames[r], _CASES[k]["n"])
        sel = _records(fl)
        trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
        tracked = _records(fl)
        sgf.KLTReplaceLostFeatures(tc, frames[r + 1], fl)
        out.append((sel, tracked, _records(fl)))
    return out


def test_public_api_from_four_threads_at_once():
    """KLTSelectGoodFeatures -> KLTTrackFeatures -> KLTReplaceLostFeatures from four threads, each with its own
    KLT_TrackingContext (different window / levels / frame size; two with the same feature count), 50 rounds concurrently, give the
    lists the same calls give on one thread.  The reference's state is per tracking context (klt.py:43-81); here every thread gets
    its own device context (backend.default_context) and every call holds that context's lock."""
    frames = [_frames_of(k, _ROUNDS) for k in range(len(_CASES))]
    want = [_api_rounds(k, frames[k], 6) for k in range(len(_CASES))]           # single thread (the main thread's context)
    got, errors = [[] for _ in _CASES], []
    gate = threading.Barrier(len(_CASES))

    def work(k):
        try:
            gate.wait(60)
            _api_rounds(k, frames[k], _ROUNDS, out=got[k])
        except BaseException as e:              # noqa: BLE001 -- re-raised by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(_CASES))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
        assert not t.is_alive()
    if errors:
        raise errors[0]
    for k in range(len(_CASES)):
        assert len(got[k]) == _ROUNDS
        assert got[k][:6] == want[k], "thread %d differs from the single-thread run" % k
        assert any(v >= 0 for _, _, v in got[k][-1][2])


def test_threads_have_their_own_device_context_and_a_tc_stays_with_its_first():
    from pyfeaturetrack_amd.backend import context_of, default_context
    sgf, trk = _api_modules()
    main_ctx = default_context()
    assert default_context() is main_ctx
    seen = {}

    def other():
        seen["ctx"] = default_context()
        seen["again"] = default_context()
        tc = make_tc(**_CASES[0]["tc"])
        f = _frames_of(0, 1)
        sgf.KLTSelectGoodFeatures(tc, f[0], 50)
        seen["tc"], seen["tc_ctx"] = tc, context_of(tc)

    t = threading.Thread(target=other)
    t.start()
    t.join(120)
    assert seen["ctx"] is seen["again"] and seen["ctx"] is not main_ctx
    assert seen["tc_ctx"] is seen["ctx"]
    assert context_of(seen["tc"]) is seen["ctx"], "a tracking context stays with the device context it was first used on"
    # ... and goes on working from this thread (its frames and pyramids live in that context's slots)
    f = _frames_of(0, 1)
    fl = sgf.KLTSelectGoodFeatures(seen["tc"], f[0], 50)
    trk.KLTTrackFeatures(seen["tc"], f[0], f[1], fl)
    ref_tc = make_tc(**_CASES[0]["tc"])
    fl2 = sgf.KLTSelectGoodFeatures(ref_tc, f[0], 50)
    trk.KLTTrackFeatures(ref_tc, f[0], f[1], fl2)
    assert _records(fl) == _records(fl2)


def test_two_threads_sharing_one_tracking_context():
    """Two threads calling the public API on ONE KLT_TrackingContext are served one call at a time (the lock of the device context the
    tracking context is bound to): every call's result is the single-thread result for its inputs."""
    sgf, trk = _api_modules()
    k = 1
    frames = _frames_of(k, 8)
    tc = make_tc(**_CASES[k]["tc"])
    n = _CASES[k]["n"]
    want = {}
    for r in range(8):
        fl = sgf.KLTSelectGoodFeatures(tc, frames[r], n)
        sel = _records(fl)
        trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
        want[r] = (sel, _records(fl))
    got, errors = {}, []

    def work(rs):
        try:
            for _ in range(5):
                for r in rs:
                    fl = sgf.KLTSelectGoodFeatures(tc, frames[r], n)
                    sel = _records(fl)
                    trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
                    got.setdefault(r, []).append((sel, _records(fl)))
        except BaseException as e:              # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(rs,)) for rs in ((0, 2, 4, 6), (1, 3, 5, 7))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
        assert not t.is_alive()
    if errors:
        raise errors[0]
    for r in range(8):
        assert len(got[r]) == 5 and all(g == want[r] for g in got[r]), "round %d" % r


def test_finalizers_never_wait_for_a_context_somebody_else_is_inside():
    """A tracking context that dies while another thread holds its device context's lock leaves its release for the next holder
    (Context._when_free / settle_deferred) instead of blocking the thread the collector happens to run on."""
    import gc
    from pyfeaturetrack_amd.backend import context_of
    sgf, _ = _api_modules()
    f = _frames_of(0, 1)
    tc = make_tc(**_CASES[0]["tc"])
    sgf.KLTSelectGoodFeatures(tc, f[0], 30)
    ctx = context_of(tc)
    base = tc._klt_slots[0]
    held, release = threading.Event(), threading.Event()

    def holder():
        with ctx.lock:
            held.set()
            release.wait(60)

    t = threading.Thread(target=holder)
    t.start()
    assert held.wait(60)
    del tc
    gc.collect()                                    # the finalizer runs here and must not block
    assert len(ctx._deferred) == 1
    release.set()
    t.join(60)
    tc2 = make_tc(**_CASES[0]["tc"])
    sgf.KLTSelectGoodFeatures(tc2, f[0], 30)        # the next call settles the deferred release and reuses the slots
    assert not ctx._deferred and tc2._klt_slots[0] == base


# ------------------------------------------------------------------------------------- optimistic frame reuse
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


def test_feature_objects_are_store_row_pairs_with_the_reference_attributes():
    """KLT_Feature: x / y / val and the affine fields read and written through the column store, Python types as the reference holds
    them (ints after selection, floats after tracking), pickles and copies; lists handed out by the API are complete
    plain-list-compatible lists."""
    import copy
    import pickle
    from pyfeaturetrack_amd.klt import KLT_Feature, shared_store
    sgf, trk = _api_modules()
    f = _frames_of(0, 1)
    tc = make_tc(**_CASES[0]["tc"])
    fl = sgf.KLTSelectGoodFeatures(tc, f[0], 60)
    assert type(fl[0]) is KLT_Feature and list.__len__(fl) == 60 and shared_store(fl) is fl._store
    assert all(type(a.x) is int and type(a.y) is int and type(a.val) is int for a in fl if a.val >= 0)
    plain = list(fl)
    trk.KLTTrackFeatures(tc, f[0], f[1], plain)                   # a plain-list copy is recognised as the same rows
    assert all(type(a.x) is float and type(a.y) is float for a in fl)
    a = fl[7]
    a.x, a.y, a.val = 5, 2.5, 3
    assert (a.x, a.y, a.val) == (5, 2.5, 3) and type(a.x) is int and fl._store.x[7] == 5.0
    b = pickle.loads(pickle.dumps(a))
    assert (b.x, b.y, b.val) == (5, 2.5, 3) and copy.deepcopy(a).y == 2.5
    lone = KLT_Feature()
    assert (lone.x, lone.y, lone.val) == (-1, -1, -1) and lone != KLT_Feature() and lone == lone
    fl.append(lone)
    assert shared_store(fl) is None                                # an edited list falls back to per-feature access
    trk.KLTTrackFeatures(tc, f[0], f[1], fl)
    assert lone.val == -1


def test_own_attributes_of_features_survive_the_klt_calls():
    """klt.py:249-263 / selectGoodFeatures.py:117-128: a KLT_Feature of the reference is an attribute bag; a script that tags its features
    (`feat.track_id = k`) finds the tags after KLTTrackFeatures / KLTReplaceLostFeatures, the calls stay on the column path, and the objects
    of a tagged list are never handed to another list."""
    import numpy as np
    from pyfeaturetrack_amd.klt import shared_store
    sgf, trk = _api_modules()
    f = _frames_of(0, 2)
    tc = make_tc(**_CASES[0]["tc"])
    n = 83                                                         # (a length no other test uses: the recycling pool is per length)
    ref = sgf.KLTSelectGoodFeatures(tc, f[0], n)
    trk.KLTTrackFeatures(tc, f[0], f[1], ref)
    want = _records(ref)
    del ref                                                        # (its objects wait in the pool)
    fl = sgf.KLTSelectGoodFeatures(tc, f[0], n)                    # ... and are taken over here
    for k, feat in enumerate(fl):
        feat.track_id = k
    fl[4].history = [(fl[4].x, fl[4].y)]
    trk.KLTTrackFeatures(tc, f[0], f[1], fl)
    assert shared_store(fl) is fl._store and _records(fl) == want
    assert [feat.track_id for feat in fl] == list(range(n)) and fl[4].history[0] == (int(fl[4].history[0][0]), int(fl[4].history[0][1]))
    sgf.KLTReplaceLostFeatures(tc, f[1], fl)
    assert [feat.track_id for feat in fl] == list(range(n))
    assert np.array(fl, dtype=object).shape == (n,)
    ids = {id(a) for a in fl}
    keep = fl[4]
    del fl, feat
    again = sgf.KLTSelectGoodFeatures(tc, f[0], n)
    assert keep.track_id == 4 and not any(hasattr(a, "track_id") for a in again) and id(keep) not in {id(a) for a in again}
    trk.KLTTrackFeatures(tc, f[0], f[1], again)
    assert _records(again) == want and len(ids) == n


@default_lists
def test_feature_objects_of_a_dropped_list_serve_the_next_selection():
    """klt._recycled through the public API: a per-frame `fl = KLTSelectGoodFeatures(...)` loop alternates between two sets of
    feature objects; a list one of whose features somebody still holds is never taken over; results are those of fresh lists."""
    from pyfeaturetrack_amd import klt
    sgf, trk = _api_modules()
    f = _frames_of(0, 2)
    tc = make_tc(**_CASES[0]["tc"])
    n = 77                                                         # (a length no other test uses: the pool is per length)
    fl = sgf.KLTSelectGoodFeatures(tc, f[0], n)
    want0 = _records(fl)
    first_ids = {id(a) for a in fl}
    trk.KLTTrackFeatures(tc, f[0], f[1], fl)
    want1 = _records(fl)
    del fl
    seen = []
    for r in range(6):
        fl = sgf.KLTSelectGoodFeatures(tc, f[0], n)
        seen.append({id(a) for a in fl})
        assert _records(fl) == want0
        trk.KLTTrackFeatures(tc, f[0], f[1], fl)
        assert _records(fl) == want1
    assert seen[0] == first_ids, "the dropped list's objects were not taken over"
    assert seen[2] == seen[0] and seen[3] == seen[1] and seen[0].isdisjoint(seen[1])
    held = fl[5]
    held_rec = (held.x, held.y, held.val)
    ids = {id(a) for a in fl}
    del fl
    fl = sgf.KLTSelectGoodFeatures(tc, f[1], n)                    # would take the objects over -- but one of them is held
    fl2 = sgf.KLTSelectGoodFeatures(tc, f[1], n)
    # (the other objects of the dropped list were freed, so their addresses may come back: only what is alive can be told apart by id)
    assert id(held) not in {id(a) for a in fl} | {id(a) for a in fl2} and held._s is not fl._store and held._s is not fl2._store
    assert len(ids) == n
    assert (held.x, held.y, held.val) == held_rec, "a feature somebody held was rewritten"
    klt.RECYCLE_FEATURE_OBJECTS, was = False, klt.RECYCLE_FEATURE_OBJECTS
    try:
        del fl, fl2
        assert _records(sgf.KLTSelectGoodFeatures(tc, f[0], n)) == want0
    finally:
        klt.RECYCLE_FEATURE_OBJECTS = was


# ------------------------------------------------------------------------------------- ADVICE r4
def test_device_free_unadopts_only_frames_inside_the_freed_allocation():
    """a slot that once held an uploaded SMALL frame, then adopted a LARGER one from a klt_device_alloc buffer, holds
    no frame after klt_device_free of that buffer (its own raw buffer holds an older, smaller image: a rebuild must not read it with the
    adopted frame's size); a slot adopted from ANOTHER allocation keeps its frame and still builds."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4))
        small = synth.synth_pair(160, 120, seed=1)[0]
        big, other = synth.synth_pair(640, 480, seed=2)
        a, b = c.device_alloc(big.nbytes), c.device_alloc(other.nbytes)
        c.device_write(a, big)
        c.device_write(b, other)
        c.upload(0, small)                                     # slot 0 owns a 160 x 120 raw buffer
        c.build_pyramids(0)
        c.adopt_u8(0, a, 640, 480)
        c.adopt_u8(1, b, 640, 480)
        c.build_pyramids_batch([0, 1], sync=True)
        want = c.download_level(1, 0, 0).copy()
        c.device_free(a)
        assert not c.frame_resident(0) and not c.pyramids_valid(0)
        with pytest.raises(KltBackendError, match="no frame"):
            c.build_pyramids(0)
        assert c.frame_resident(1), "a slot adopted from other memory lost its frame"
        c.build_pyramids(1)
        assert np.array_equal(c.download_level(1, 0, 0), want)
        c.upload(0, small)                                     # the slot is usable again
        c.build_pyramids(0)
        assert c.level_dims(0, 0) == (160, 120)
    finally:
        c.close()


def test_uploads_into_one_slot_without_a_build_in_between_stay_ordered():
    """consecutive klt_upload_u8_async calls go round-robin over the copy streams; a slot uploaded again and again
    without a build in between (each copy then lands in a raw buffer an earlier copy -- on another stream -- was written to) always
    ends up holding the LAST frame sent."""
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4))
        w, h = 1920, 1080
        frames = [np.full((h, w), 10 * k + 5, np.uint8) for k in range(7)]
        pins = []
        for f in frames:
            p = c.pinned_array((h, w))
            p[...] = f
            pins.append(p)
        ref = Context(0)
        try:
            ref.configure(make_tc(levels=2, ss=4))
            for rounds in (2, 3, 4, 5, 7):
                for k in range(rounds):
                    c.upload_async(0, pins[k])
                c.build_pyramids(0)
                ref.upload(0, frames[rounds - 1])
                ref.build_pyramids(0)
                assert np.array_equal(c.download_level(0, 0, 0), ref.download_level(0, 0, 0)), "after %d uploads" % rounds
        finally:
            ref.close()
    finally:
        c.close()


def test_prepared_replacement_vs_oracle_on_random_draws():
    """tests/fuzz/fuzz_parity.py --prepared: klt_select_prepare_async (row pass + cols_eigen_pipe) followed by
    klt_select_begin_async / klt_select_finish with REPLACING_SOME, on frames with more than 262144 candidates (the prefilter cut and
    the four-tiles-per-workgroup passes behind it run) -- every record identical to the oracle's.  250 draws ran in
    tests/fuzz/fuzz_long.sh; 6 stay in the suite."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from pyfeaturetrack_amd.backend import Context
    rng = np.random.default_rng(5)
    c = Context(0)
    try:
        done = 0
        while done < 6:
            t = fz.draw(rng, 600000, 2500, 1000)
            if t["w"] * t["h"] < 330000:
                continue
            bad = fz.run_prepared_trial(c, t)
            assert bad is None, "draw %d: %s differs from the oracle: %r" % (done, bad, t)
            done += 1
    finally:
        c.close()


# ------------------------------------------------------------------------------------- the reference's helper functions by name
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


# ------------------------------------------------------------------------------------- opt-in tree sums in the tracker
def _tree_vs_golden(ctx, fl, want_x, want_y, want_val, what, s1=0, s2=1):
    """tracker with KLT_OPT_TRACK_TREE_SUMS on the given list: identical status words, positions within the north star's 1e-3 px of the
    reference's (the default kernel gives them bit for bit); returns (max |d|, positions that are not bit-identical)"""
    ctx.set_option(18, 1)
    try:
        out, _ = ctx.track(s1, s2, fl)
    finally:
        ctx.set_option(18, 0)
    assert np.array_equal(out["val"].astype(np.int64), np.asarray(want_val).astype(np.int64)), \
        "%s: %d status words differ" % (what, int((out["val"] != want_val).sum()))
    ok = out["val"] == 0
    dx = np.abs(out["x"][ok].astype(np.float64) - want_x[ok])
    dy = np.abs(out["y"][ok].astype(np.float64) - want_y[ok])
    err = float(max(dx.max(), dy.max())) if ok.any() else 0.0
    assert err <= 1e-3, "%s: %g px" % (what, err)
    return err, int(((dx != 0) | (dy != 0)).sum())


def test_tree_sums_tracker_vs_reference_goldens(golden_dir, cfg1, img0, img1):
    """KLT_OPT_TRACK_TREE_SUMS (north_star: "gradient-sum / SSD reduction with wave-level shuffles", outputs within
    1e-3 px): the five window sums and the residue reduced by a DPP butterfly in registers -- same precision as the reference
    (trackFeaturesUtils.pyx:246-305), other order of the additions.  Per call on the reference's own inputs (its selected lists), not
    chained: status words identical and |dx|, |dy| <= 1e-3 px against the reference's tracked lists at cfg-1 / 2 / 3 / 4 / 5 size and on the
    220 random draws it ran.  The default stays the bit-exact kernel."""
    import os
    from helpers import baseline_case, random_draws
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE
    big = np.load(os.path.join(golden_dir, "baseline_sizes.npz"))
    c = Context(0)
    worst, inexact, total = 0.0, 0, 0
    try:
        for tag, mr in (("r10", 10.0), ("rnone", None)):                      # cfg-1: img0 -> img1, 100 features (one feature per wavefront:
            c.configure(make_tc(max_residue=mr))                            # the option does not apply, the records are the exact ones)
            c.upload(0, img0)
            c.upload(1, img1)
            c.build_pyramids_batch([0, 1], sync=True)
            fl, _ = c.select(0, 100)
            e, k = _tree_vs_golden(c, fl, cfg1["trk100_%s_x" % tag], cfg1["trk100_%s_y" % tag], cfg1["trk100_%s_val" % tag], "cfg-1 " + tag)
            assert (e, k) == (0.0, 0)
        for tag in ("cfg2", "cfg4", "cfg3", "cfg5"):
            frames, tc, n = baseline_case(tag)
            c.configure(tc)
            c.upload(0, frames[0])
            c.upload(1, frames[1])
            c.build_pyramids_batch([0, 1], sync=True)
            fl = np.zeros(n, FEAT_DTYPE)
            fl["x"], fl["y"], fl["val"] = big[tag + "_sel_x"], big[tag + "_sel_y"], big[tag + "_sel_val"]
            e, k = _tree_vs_golden(c, fl, big[tag + "_trk_x"], big[tag + "_trk_y"], big[tag + "_trk_val"], tag)
            worst, inexact, total = max(worst, e), inexact + k, total + n
        assert inexact > 0, "the tree sums gave the reference's bits everywhere: is the option wired?"
        for name in ("random_draws.npz", "random_draws_large.npz"):
            for t, tc, f0, f1, want in random_draws(golden_dir, name):
                c.configure(tc)
                c.upload(0, f0)
                c.upload(1, f1)
                c.build_pyramids_batch([0, 1], sync=True)
                fl = np.zeros(t["n"], FEAT_DTYPE)
                fl["x"], fl["y"], fl["val"] = want["sel"]
                e, k = _tree_vs_golden(c, fl, *want["trk"], what="draw %r" % (t["seed"],))
                worst = max(worst, e)
    finally:
        c.close()
    print("tree sums: worst |d| = %g px, %d of %d positions at the BASELINE sizes not bit-identical" % (worst, inexact, total))


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


def test_feature_buffers_mapped_into_pinned_host_memory():
    """klt_featbuf_map_host: the tracker reads and writes pinned host records in place -- same records as through device buffers and two
    copies; the ordinary copies still work on a mapped buffer; pageable memory is refused; unmapping empties the buffer; the Python
    layer unmaps before it frees the pinned arrays of a list length it evicts."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4, max_residue=10.0))
        f0, f1 = synth.synth_pair(640, 480, seed=9)
        c.upload(0, f0)
        c.upload(1, f1)
        c.build_pyramids_batch([0, 1], sync=True)
        n = 300
        fl, _ = c.select(0, n)
        want, _ = c.track(0, 1, fl)                                   # device buffers, synchronous copies
        rin, rout = c.pinned_array((n,), FEAT_DTYPE), c.pinned_array((n,), FEAT_DTYPE)
        rin[...] = fl
        rout["val"] = 77
        c._check(c._lib.klt_featbuf_map_host(c._h, 40, rin.ctypes.data, n))
        c._check(c._lib.klt_featbuf_map_host(c._h, 41, rout.ctypes.data, n))
        c.track_async(0, 1, 40, 41, n)
        c.sync()
        assert rout.tobytes() == want.tobytes(), "records written in place differ from the copied ones"
        assert c.featbuf_download(41, n).tobytes() == want.tobytes()            # an ordinary download of a mapped buffer
        c.featbuf_upload(40, want)                                              # ... and an upload into one: lands in the host array
        assert rin.tobytes() == want.tobytes()
        with pytest.raises(KltBackendError, match="pinned"):
            c._check(c._lib.klt_featbuf_map_host(c._h, 42, np.zeros(n, FEAT_DTYPE).ctypes.data, n))
        c._check(c._lib.klt_featbuf_map_host(c._h, 41, None, 0))                # unmapped: empty
        with pytest.raises(KltBackendError):
            c.featbuf_download(41, n)
        c.track_async(0, 1, 40, 41, n)                                          # ... and usable as an ordinary device buffer again
        assert c.featbuf_download(41, n)["val"].tolist() == c.track(0, 1, want)[0]["val"].tolist()
        # the API's own mapping: ONE pair of pinned arrays serves every list length (views), so changing the length does not remap
        for k in range(9):
            m = 50 + k
            c.host_records(m)[0][...] = fl[:m]
            c.track_enqueue(0, 1, m)
            got = c.track_complete(m)
            assert got.tobytes() == want[:m].tobytes(), m
        first_map = c._mapped_records
        assert first_map is not None and len(c._host_records[0]) >= 58
        # ... it grows with the longest list (unmapped, freed, allocated anew, mapped again) ...
        assert n > len(c._host_records[0])
        c.host_records(n)[0][...] = fl
        c.track_enqueue(0, 1, n)
        assert c.track_complete(n).tobytes() == want.tobytes()
        grown = c._mapped_records
        assert grown != first_map and len(c._host_records[0]) >= n
        # ... and a script that alternates between two list lengths keeps that one mapping (two device-wide waits per call)
        for k in range(6):
            m = (57, n)[k % 2]
            c.host_records(m)[0][...] = fl[:m]
            c.track_enqueue(0, 1, m)
            assert c.track_complete(m).tobytes() == want[:m].tobytes(), m
            assert c._mapped_records is grown
    finally:
        c.close()


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
    its walk reaches the point, selectGoodFeatures.py:90-91); a zero key still ends the list; the context is usable afterwards."""
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
            out, placed = c.min_distance_walk(good[:2] + [0, key(6.0, 1, 1)], ncols, nrows, mindist, True, fl)      # a zero key ends the list
            assert placed == 2
            out, placed = c.min_distance_walk(good, ncols, nrows, mindist, True, fl)
            assert placed == 3
    finally:
        c.close()


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
    objects, scores prepared ahead: none of it may show).  8400 draws ran when the mode was written (profiles/r05_fuzz_seeds.txt); 25
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
