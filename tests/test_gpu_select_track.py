"""Selection and tracking beyond the golden cases of test_gpu_parity.py (SURVEY 8 a-9 ... a-23, N-1): cfg-3 at its full size with the
affine check on (all three affine models), the tracker enqueued ahead of a replacement's look, replacement from prepared scores on
random draws, degenerate frames, the opt-in butterfly-sum tracker against the reference goldens.  (Folded by component from the
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


W, H, NF = 1920, 1080, 5000


STATE_FIELDS = ("valid", "aff_x", "aff_y", "Axx", "Ayx", "Axy", "Ayy", "pad")


def cfg3_tc(mode):
    import bench
    tc = bench.cfg3_context()                      # 15x15, 4 levels / ss 2 (border 108), affine window 15x15
    tc.affineConsistencyCheck = mode
    return tc


def shifted_frames():
    import bench
    return bench.cfg3_frames(4)                    # what `bench.py --config cfg3` times: pure translation by (1.1, -0.7) per frame


def warped_frames():
    """a small similarity + shear per frame on top of the translation: the affine matrices have something to converge to"""
    base = synth.synth_base(W, H, 1)
    A_step = np.array([[1.0015, 0.0012], [-0.0009, 0.9988]])
    frames, A = [], np.eye(2)
    for k in range(4):
        frames.append(synth.warp_frame(base, A, (1.1 * k, -0.7 * k)))
        A = A_step @ A
    return frames


def three_calls_gpu(frames, tc):
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        c.configure(tc)
        for k, f in enumerate(frames):
            c.upload(k, f)
        c.build_pyramids_batch(list(range(len(frames))), sync=True)
        fl, placed = c.select(0, NF, use_pyramid=True)
        assert placed == NF
        c.affine_alloc(0, NF)
        c.featbuf_upload(0, fl)
        hist = [(fl.copy(), None)]
        for k in range(1, len(frames)):
            c.track_affine_async(k - 1, k, k - 1, k, NF, 0)          # through the asynchronous ABI entry point bench.py times
            hist.append((c.featbuf_download(k, NF), c.affine_download(0, NF)))
        return hist
    finally:
        c.close()


def three_calls_oracle(frames, tc):
    from oracle import klt_oracle as ko
    p, ap = params_from_tc(tc), affine_params_from_tc(tc)
    ko.set_threads(min(16, os.cpu_count() or 1))
    try:
        f32 = [f.astype(np.float32) for f in frames]
        fl = ko.select_good_features(p, f32[0], NF)
        st = ko.AffineState(ap, NF)
        P = [ko.Pyramids(p, f) for f in f32]
        hist = [(fl.copy(), None)]
        for k in range(1, len(frames)):
            ko.track_features_affine(p, P[k - 1], P[k], fl, st)
            hist.append((fl.copy(), st.rec.copy()))
        return hist
    finally:
        ko.set_threads(1)


def assert_same_history(g, o, what):
    for k, ((gfl, grec), (ofl, orec)) in enumerate(zip(g, o)):
        assert np.array_equal(gfl["val"], ofl["val"]), "%s, call %d: %d status codes differ" % (what, k, int((gfl["val"] != ofl["val"]).sum()))
        assert np.array_equal(gfl["x"], ofl["x"]) and np.array_equal(gfl["y"], ofl["y"]), "%s, call %d: positions" % (what, k)
        if grec is not None:
            for name in STATE_FIELDS:
                assert np.array_equal(grec[name], orec[name]), "%s, call %d: affine state field %s" % (what, k, name)


@pytest.mark.parametrize("mode,frames_of", [(2, "shifted"), (2, "warped"), (1, "warped"), (0, "warped")])
def test_cfg3_full_size_with_the_affine_check_on(mode, frames_of):
    """BASELINE cfg-3 at its real geometry -- 1920x1080, 15x15 window, 4 levels / ss 2, 5000 features, four frames = three
    KLTTrackFeatures calls (the first stores the templates, the second and third run the check; interface:
    /root/reference trackFeatures.py:347-399) -- HIP == oracle on val, x, y and on valid, aff_x / aff_y, the four entries of A and
    the iteration count of every feature, after every call.  Parity of the check itself is UNPINNED (the reference does not define
    the functions it calls there); this is the implementation against the stated specification."""
    frames = shifted_frames() if frames_of == "shifted" else warped_frames()
    tc = cfg3_tc(mode)
    g, o = three_calls_gpu(frames, tc), three_calls_oracle(frames, tc)
    assert_same_history(g, o, "mode %d, %s frames" % (mode, frames_of))
    last_fl, last_rec = g[-1]
    assert (last_fl["val"] == 0).sum() > 0.9 * NF
    live = last_fl["val"] == 0
    assert last_rec["pad"][live].min() >= 1, "the check ran on every surviving feature"
    if frames_of == "warped" and mode == 2:
        want = np.linalg.matrix_power(np.array([[1.0015, 0.0012], [-0.0009, 0.9988]]), 3)
        got = np.array([[np.median(last_rec["Axx"][live]), np.median(last_rec["Axy"][live])],
                        [np.median(last_rec["Ayx"][live]), np.median(last_rec["Ayy"][live])]])
        assert np.abs(got - want).max() < 4e-3, (got, want)


def test_tracker_enqueued_before_the_selections_look_is_repeated_when_the_list_changes():
    """klt_select_finish returns 1 when the host's look made the selection rewrite the list after the first half's launches had run
    (here: a score ramp -- one dependency chain across the frame -- needs far more minimum-distance passes than a fresh context
    enqueues before it looks); a tracker launched in between has then read an unfinished list and must be launched again, which is
    what KLTTrackSequence and bench.py --config cfg5 do.  The repeated launch gives the records of the plain order (select, then
    track), whether or not the look asked for more (real scores: either way)."""
    from helpers import make_tc
    from pyfeaturetrack_amd.backend import Context, SELECTING_ALL
    n, w, h = 400, 500, 300
    base = synth.synth_base(w, h, 13)
    f0, f1 = synth.synth_frame(w, h, 13, 0, shift=(1.5, -1.0), base=base), synth.synth_frame(w, h, 13, 1, shift=(1.5, -1.0), base=base)
    tc = make_tc(levels=2, ss=2, mindist=10)
    p = params_from_tc(tc)
    bx, by = int(max(p.borderx, p.window_width / 2.0)), int(max(p.bordery, p.window_height / 2.0))
    ys, xs = np.mgrid[0:h - 2 * by, 0:w - 2 * bx]
    ramp = (10.0 + xs + 0.001 * ys).astype(np.float32)
    for scores in (ramp, None):
        c = Context(0)                                         # fresh: it enqueues its default number of passes before the first look
        try:
            c.configure(tc)
            c.upload(0, f0)
            c.upload(1, f1)
            c.build_pyramids_batch([0, 1], sync=True)
            # plain order
            if scores is not None:
                c.set_score_override(scores)
            c.select_async(0, SELECTING_ALL, True, 0, n)
            c.track_async(0, 1, 0, 1, n)
            want_list, want_trk = c.featbuf_download(0, n), c.featbuf_download(1, n)
            assert (want_list["val"] > 0).sum() > n // 2
        finally:
            c.close()
        c = Context(0)
        try:
            c.configure(tc)
            c.upload(0, f0)
            c.upload(1, f1)
            c.build_pyramids_batch([0, 1], sync=True)
            if scores is not None:
                c.set_score_override(scores)
            c.select_begin(0, SELECTING_ALL, True, 0, n)
            c.track_async(0, 1, 0, 2, n)                       # reads the list the selection may still be working on
            rewritten = c.select_finish()
            assert rewritten or scores is None, "the ramp needs more passes than a fresh context enqueues before it looks"
            if rewritten:
                c.track_async(0, 1, 0, 2, n)
            assert np.array_equal(c.featbuf_download(0, n), want_list)
            assert np.array_equal(c.featbuf_download(2, n), want_trk)
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
