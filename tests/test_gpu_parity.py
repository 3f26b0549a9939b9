"""Parity of the HIP path (through the C ABI) with the oracle and with the reference's goldens.

Integer / index results and every f32 image are compared bit-for-bit; tracked positions are
compared bit-for-bit as well (the north-star tolerance is 1e-3 px -- TOL below -- but the kernels
reproduce the reference's arithmetic order, so the stricter check is the regression guard).
"""
import os

import numpy as np
import pytest

from helpers import make_tc, params_from_tc, synth251_frames, sha_bytes

pytestmark = pytest.mark.gpu
TOL = 1e-3   # px, BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ko():
    from oracle import klt_oracle
    return klt_oracle


def assert_same(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    bad = np.flatnonzero(a.ravel() != b.ravel())
    if bad.size:
        i = bad[0]
        raise AssertionError("%s: %d of %d differ; first at flat index %d: %r vs %r" %
                             (what, bad.size, a.size, i, a.ravel()[i], b.ravel()[i]))


def assert_feats(fl, gx, gy, gv, what):
    assert_same(fl["val"].astype(np.int64), np.asarray(gv, np.int64), what + ".val")
    dx = np.abs(fl["x"].astype(np.float64) - np.asarray(gx, np.float64)).max()
    dy = np.abs(fl["y"].astype(np.float64) - np.asarray(gy, np.float64)).max()
    assert dx <= TOL and dy <= TOL, "%s: position error %g / %g px exceeds %g" % (what, dx, dy, TOL)
    assert_same(fl["x"].astype(np.float64), gx, what + ".x (bit-exact)")
    assert_same(fl["y"].astype(np.float64), gy, what + ".y (bit-exact)")


def oracle_feats(fl):
    return fl["x"].astype(np.float64), fl["y"].astype(np.float64), fl["val"].astype(np.int64)


# ------------------------------------------------------------------------------ convolutions
def test_smooth_and_gradients_img0(ctx, ko, cfg1, img0):
    from pyfeaturetrack_amd.convolve import _computeKernels
    g07, _ = _computeKernels(0.1 * 7)     # smooth_sigma_fact * max(window) as the reference computes it
    sm = ctx.smooth(img0.astype(np.float32), g07)
    assert_same(sm, cfg1["sel_smooth"], "smooth(img0)")
    g, d = _computeKernels(1.0)
    gx, gy = ctx.gradients(sm, g, d)
    assert_same(gx, cfg1["sel_gx"], "gradx(img0)")
    assert_same(gy, cfg1["sel_gy"], "grady(img0)")


@pytest.mark.parametrize("shape,sigma", [((187, 251), 3.6), ((67, 120), 7.2), ((9, 7), 7.2), ((5, 300), 1.8),
                                         ((300, 3), 1.5), ((1, 1), 1.0), ((64, 64), 0.7), ((130, 257), 1.0)])
def test_convolutions_vs_oracle(ctx, ko, shape, sigma):
    from pyfeaturetrack_amd.convolve import _computeKernels
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = (rng.random(shape) * 255).astype(np.float32)
    g, d = _computeKernels(sigma)
    assert_same(ctx.smooth(img, g), ko.smooth(img, sigma), "smooth %s sigma %s" % (shape, sigma))
    gx, gy = ctx.gradients(img, g, d)
    ogx, ogy = ko.gradients(img, sigma)
    assert_same(gx, ogx, "gradx %s sigma %s" % (shape, sigma))
    assert_same(gy, ogy, "grady %s sigma %s" % (shape, sigma))


# ---------------------------------------------------------------------------------- pyramids
def test_pyramids_cfg1(ctx, cfg1, img0, img1):
    ctx.configure(make_tc())
    for slot, (name, im) in enumerate((("p0", img0), ("p1", img1))):
        ctx.upload(slot, im)
        ctx.build_pyramids(slot)
        for l in range(2):
            for pi, w in enumerate(("img", "gx", "gy")):
                assert_same(ctx.download_level(slot, pi, l), cfg1["%s_%s_%d" % (name, w, l)], "%s %s level %d" % (name, w, l))


def test_pyramids_f32_upload_equals_u8(ctx, img0):
    ctx.configure(make_tc())
    ctx.upload(0, img0)
    ctx.build_pyramids(0)
    ctx.upload(1, img0.astype(np.float32))
    ctx.build_pyramids(1)
    for l in range(2):
        for pi in range(3):
            assert_same(ctx.download_level(0, pi, l), ctx.download_level(1, pi, l), "u8 vs f32 upload")


def test_pyramids_synth251(ctx, ko, synth251):
    fr = synth251_frames()
    tc = make_tc(levels=3, ss=2, max_residue=10.0)
    ctx.configure(tc)
    p = params_from_tc(tc)
    for slot, name in ((0, "p0"), (1, "p1")):
        ctx.upload(slot, fr[slot])
        ctx.build_pyramids(slot)
        P = ko.Pyramids(p, fr[slot].astype(np.float32))
        assert [ctx.level_dims(slot, l) for l in range(3)] == [(251, 187), (125, 93), (62, 46)]
        for l in range(3):
            for pi, w in enumerate(("img", "gx", "gy")):
                got = ctx.download_level(slot, pi, l)
                assert_same(got, P.level(w, l), "synth %s %s level %d vs oracle" % (name, w, l))
                assert_same(sha_bytes(got), synth251["%s_%s_%d_sha" % (name, w, l)], "synth %s %s level %d sha" % (name, w, l))


def test_generic_path_and_batched_build(ctx, ko, cfg1, img0, img1):
    """KLT_OPT_FUSED_KERNELS=0 (generic two-pass kernels) and the batched build give the same pyramids."""
    ctx.configure(make_tc())
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.upload(2, img1.astype(np.float32))
    try:
        ctx.set_option(1, 0)
        ctx.build_pyramids(0)
        ctx.build_pyramids(1)
        for slot, name in ((0, "p0"), (1, "p1")):
            for l in range(2):
                for pi, w in enumerate(("img", "gx", "gy")):
                    assert_same(ctx.download_level(slot, pi, l), cfg1["%s_%s_%d" % (name, w, l)], "generic %s %s %d" % (name, w, l))
    finally:
        ctx.set_option(1, 1)
    ctx.build_pyramids_batch([0, 1, 2], sync=True)      # u8, u8, f32 frames: two launch groups
    for slot, name in ((0, "p0"), (1, "p1"), (2, "p1")):
        for l in range(2):
            for pi, w in enumerate(("img", "gx", "gy")):
                assert_same(ctx.download_level(slot, pi, l), cfg1["%s_%s_%d" % (name, w, l)], "batched %s %s %d" % (name, w, l))


@pytest.mark.parametrize("attrs,window,levels,ss", [({"grad_sigma": 1.5}, 7, 2, 4), ({"grad_sigma": 0.7}, 7, 3, 2),
                                                     ({"smooth_sigma_fact": 0.3}, 7, 2, 2), ({}, 21, 2, 4),
                                                     ({"pyramid_sigma_fact": 0.6}, 7, 3, 4), ({"grad_sigma": 2.2}, 9, 2, 8)])
def test_unusual_sigmas_take_the_runtime_sized_kernels(ctx, ko, attrs, window, levels, ss):
    """Tap counts other than 5 / 21 / 7+7 fall back to the runtime-sized LDS kernels (or the generic two-pass ones);
    results stay bit-identical to the oracle, selection and tracking included."""
    from pyfeaturetrack_amd import synth
    base = synth.synth_base(300, 220, 21)
    f0, f1 = synth.shift_frame(base, 0, 0), synth.shift_frame(base, 1.4, 0.9)
    tc = make_tc(levels=levels, ss=ss, window=window, max_residue=20.0, **attrs)
    tc.KLTUpdateTCBorder()
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids_batch([0, 1])
    P0, P1 = ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32))
    for l in range(levels):
        for pi, w in enumerate(("img", "gx", "gy")):
            assert_same(ctx.download_level(1, pi, l), P1.level(w, l), "%s level %d %s" % (attrs, l, w))
    if 2 * tc.borderx + 20 < 300 and 2 * tc.bordery + 20 < 220:
        fl, _ = ctx.select(0, 30, use_pyramid=True)
        ofl = ko.select_good_features(p, f0.astype(np.float32), 30)
        assert_feats(fl, *oracle_feats(ofl), what="%s select" % attrs)
        out, _ = ctx.track(0, 1, fl)
        ko.track_features(p, P0, P1, ofl)
        assert_feats(out, *oracle_feats(ofl), what="%s track" % attrs)


@pytest.mark.parametrize("shape,f32_input", [((1080, 1920), False), ((1080, 1920), True), ((1013, 1250), False), ((2160, 3840), False),
                                             ((1100, 1001), False)])
def test_fused_first_reduction_on_and_off(ctx, ko, shape, f32_input):
    """KLT_OPT_FUSED_HREDUCE: level 1 from the H1 planes written by the level-0 kernel + the vertical-pass kernel (default for
    subsampling 4 on large frames), or from the separate reduction kernel: both equal the oracle's pyramids, every level."""
    from pyfeaturetrack_amd import synth
    img = synth.synth_frame(shape[1], shape[0], 5, 0)
    tc = make_tc(levels=3, ss=4)
    ctx.configure(tc)
    P = ko.Pyramids(params_from_tc(tc), img.astype(np.float32))
    for fused in (1, 0):
        try:
            ctx.set_option(12, fused)
            ctx.upload(0, img.astype(np.float32) if f32_input else img)
            ctx.build_pyramids(0)
            for l in range(3):
                for pi, w in enumerate(("img", "gx", "gy")):
                    assert_same(ctx.download_level(0, pi, l), P.level(w, l), "fused reduction %d, %s level %d, %dx%d" % (fused, w, l, shape[1], shape[0]))
        finally:
            ctx.set_option(12, 1)


@pytest.mark.parametrize("window,levels,ss,shape", [(7, 3, 8, (700, 900)), (15, 3, 2, (301, 447)), (5, 2, 4, (64, 64)),
                                                    (7, 4, 2, (123, 77)), (7, 2, 8, (40, 50)), (9, 2, 4, (17, 333))])
def test_pyramids_various_geometries_vs_oracle(ctx, ko, window, levels, ss, shape):
    rng = np.random.default_rng(window * 100 + ss)
    img = (rng.random(shape) * 255).astype(np.uint8)
    tc = make_tc(levels=levels, ss=ss, window=window)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, img)
    ctx.build_pyramids(0)
    P = ko.Pyramids(p, img.astype(np.float32))
    for l in range(levels):
        for pi, w in enumerate(("img", "gx", "gy")):
            assert_same(ctx.download_level(0, pi, l), P.level(w, l), "w%d L%d ss%d %s level %d" % (window, levels, ss, w, l))


# --------------------------------------------------------------------------------- selection
def test_select_internals_cfg1(ctx, cfg1, img0):
    ctx.configure(make_tc())
    ctx.upload(2, img0)
    fl, placed = ctx.select(2, 100)
    assert placed == 100
    assert_same(ctx.select_intermediate(0), cfg1["sel_smooth"], "selection smoothed image")
    assert_same(ctx.select_intermediate(1), cfg1["sel_gx"], "selection gradx")
    assert_same(ctx.select_intermediate(2), cfg1["sel_gy"], "selection grady")
    assert_same(ctx.select_intermediate(3), cfg1["sel_val"], "eigenvalue map")
    assert_feats(fl, cfg1["sel100_x"], cfg1["sel100_y"], cfg1["sel100_val"], "select 100 (parallel minimum distance)")
    ctx.set_option(8, 0)                          # the sorted serial walk keeps the full sorted candidate list
    try:
        fl, placed = ctx.select(2, 100)
    finally:
        ctx.set_option(8, 1)
    assert placed == 100
    val, x, y = ctx.sorted_candidates(20000)
    n = len(val)
    gv = cfg1["sel_sorted_val"]
    n_ref = int(np.count_nonzero(gv >= 1.0))      # the device list drops val < max(min_eigenvalue, 1)
    assert n == min(n_ref, 20000) or n_ref == 20000
    assert_same(val, gv[:n], "sorted candidate values")
    assert_same(x, cfg1["sel_sorted_x"][:n], "sorted candidate x")
    assert_same(y, cfg1["sel_sorted_y"][:n], "sorted candidate y")
    assert_feats(fl, cfg1["sel100_x"], cfg1["sel100_y"], cfg1["sel100_val"], "select 100")


@pytest.mark.parametrize("n", [50, 100, 300])
def test_select_cfg1(ctx, cfg1, img0, n):
    ctx.configure(make_tc())
    ctx.upload(2, img0)
    fl, placed = ctx.select(2, n)
    assert placed == n
    assert_feats(fl, cfg1["sel%d_x" % n], cfg1["sel%d_y" % n], cfg1["sel%d_val" % n], "select %d" % n)


@pytest.mark.parametrize("shape", [(1080, 1920), (240, 320), (187, 251), (133, 260), (64, 68)])
def test_sat_variants_agree_with_oracle(ctx, ko, shape):
    """Summed-area tables: the step-synchronous wavefront pipelines (default; frames with ncols % 4 == 0) and the
    barrier-coupled kernels (KLT_OPT_SAT_VARIANT = 0, also the fallback for other widths) give the oracle's eigenvalue map."""
    from pyfeaturetrack_amd import synth
    img = synth.synth_frame(shape[1], shape[0], 3, 0)
    tc = make_tc(levels=2, ss=2) if min(shape) < 200 else make_tc()
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, img)
    n = 50
    ofl, oval = ko.select_good_features(p, img.astype(np.float32), n, want_val=True)
    for variant in (1, 0):
        try:
            ctx.set_option(10, variant)
            fl, placed = ctx.select(0, n)
            assert_same(ctx.select_intermediate(3), oval, "eigenvalue map, SAT variant %d, %dx%d" % (variant, shape[1], shape[0]))
            assert_feats(fl, *oracle_feats(ofl), what="selection, SAT variant %d" % variant)
        finally:
            ctx.set_option(10, 1)


def test_select_skip_mindist_nosmooth(ctx, cfg1, img0):
    ctx.configure(make_tc(nSkippedPixels=2, mindist=15, smoothBeforeSelecting=False))
    ctx.upload(2, img0)
    fl, _ = ctx.select(2, 60)
    assert_feats(fl, cfg1["selskip_x"], cfg1["selskip_y"], cfg1["selskip_val"], "select skip=2 mindist=15 nosmooth")
    ctx.upload(2, img0.astype(np.float32))       # f32 frame without pre-smoothing
    fl, _ = ctx.select(2, 60)
    assert_feats(fl, cfg1["selskip_x"], cfg1["selskip_y"], cfg1["selskip_val"], "select (f32 upload)")


@pytest.mark.parametrize("mindist", [0, 1, 2, 10, 40])
def test_select_mindist_and_exhaustion_vs_oracle(ctx, ko, img0, mindist):
    """more features requested than can be placed -> remaining slots are (-1,-1,KLT_NOT_FOUND)"""
    tc = make_tc(mindist=mindist)
    ctx.configure(tc)
    ctx.upload(2, img0)
    n = 3000 if mindist >= 2 else 500
    fl, placed = ctx.select(2, n)
    ofl = ko.select_good_features(params_from_tc(tc), img0.astype(np.float32), n)
    assert placed == int(np.count_nonzero(ofl["val"] >= 0))
    assert_feats(fl, *oracle_feats(ofl), what="select mindist=%d n=%d" % (mindist, n))


def test_replacing_some_cfg1(ctx, cfg1, img1):
    ctx.configure(make_tc(max_residue=10.0))
    from pyfeaturetrack_amd.backend import FEAT_DTYPE, REPLACING_SOME
    fl = np.zeros(100, FEAT_DTYPE)
    fl["x"], fl["y"], fl["val"] = cfg1["repl_in_x"], cfg1["repl_in_y"], cfg1["repl_in_val"]
    ctx.upload(2, img1)
    out, placed = ctx.select(2, 100, mode=REPLACING_SOME, fl=fl)
    assert placed == int(np.count_nonzero(cfg1["repl_in_val"] < 0))
    assert_feats(out, cfg1["repl_out_x"], cfg1["repl_out_y"], cfg1["repl_out_val"], "replace lost features")


# ---------------------------------------------------------------------------------- tracking
@pytest.mark.parametrize("tag,mr", [("r10", 10.0), ("rnone", None)])
def test_track_cfg1(ctx, cfg1, img0, img1, tag, mr):
    ctx.configure(make_tc(max_residue=mr))
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids(0, sync=False)
    ctx.build_pyramids(1, sync=False)
    fl, _ = ctx.select(0, 100, use_pyramid=False)
    ctx.track_stats_reset()
    out, k = ctx.track(0, 1, fl)
    assert_feats(out, cfg1["trk100_%s_x" % tag], cfg1["trk100_%s_y" % tag], cfg1["trk100_%s_val" % tag], "track " + tag)
    assert k == int(np.count_nonzero(cfg1["trk100_%s_val" % tag] >= 0))
    rec = cfg1["trk100_%s_iter" % tag]          # one row per trackFeatureIterateCKLT call: ..., ncols, ..., iterations
    st = ctx.track_stats()
    assert st["features"] == 100
    for lvl, nc in ((0, 320), (1, 80)):
        rows = rec[rec[:, 2] == nc]
        assert st["level_visits"][lvl] == len(rows)
        assert st["iterations"][lvl] == int(rows[:, 6].sum())


def test_empty_and_degenerate_inputs(ctx, ko, img0, img1):
    """empty feature list, a list with only lost features, a frame too small to hold any candidate"""
    from pyfeaturetrack_amd.backend import FEAT_DTYPE
    tc = make_tc(max_residue=10.0)
    ctx.configure(tc)
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids_batch([0, 1])
    out, k = ctx.track(0, 1, np.zeros(0, FEAT_DTYPE))
    assert len(out) == 0 and k == 0
    lost = np.zeros(7, FEAT_DTYPE)
    lost["val"] = [-1, -2, -3, -4, -5, -1, -4]
    lost["x"] = lost["y"] = -1
    out, k = ctx.track(0, 1, lost)
    assert k == 0 and np.array_equal(out[["x", "y", "val"]], lost[["x", "y", "val"]])
    small = img0[:50, :56].copy()                     # border 30 on every side leaves no interior pixel
    ctx.upload(2, small)
    fl, placed = ctx.select(2, 10)
    assert placed == 0 and np.all(fl["val"] == -1) and np.all(fl["x"] == -1)
    ofl = ko.select_good_features(params_from_tc(tc), small.astype(np.float32), 10)
    assert np.array_equal(fl["val"], ofl["val"])
    strip = img0[:61, :].copy()                       # exactly one candidate row
    ctx.upload(2, strip)
    fl, placed = ctx.select(2, 40)
    ofl = ko.select_good_features(params_from_tc(tc), strip.astype(np.float32), 40)
    assert_feats(fl, *oracle_feats(ofl), what="one-row candidate strip")


def test_lost_features_pass_through(ctx, cfg1, img0, img1):
    """features with val < 0 are not tracked and come back untouched (trackFeatures.py:253)"""
    ctx.configure(make_tc(max_residue=10.0))
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids_batch([0, 1])
    fl, _ = ctx.select(0, 100, use_pyramid=True)
    out, _ = ctx.track(0, 1, fl)
    fl2 = fl.copy()
    fl2["val"][::3] = -4
    fl2["x"][::3] = -1
    out2, _ = ctx.track(0, 1, fl2)
    keep = fl2["val"] >= 0
    assert np.array_equal(out2[["x", "y", "val"]][~keep], fl2[["x", "y", "val"]][~keep])
    assert np.array_equal(out2["x"][keep], out["x"][keep]) and np.array_equal(out2["val"][keep], out["val"][keep])


def test_track_retain(ctx, cfg1, img0, img1):
    ctx.configure(make_tc(max_residue=10.0, retainTrackers=True))
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, 100)
    out, _ = ctx.track(0, 1, fl)
    assert_feats(out, cfg1["trk100_retain_x"], cfg1["trk100_retain_y"], cfg1["trk100_retain_val"], "track retainTrackers")


@pytest.mark.parametrize("window,levels,ss,retain,mr", [(7, 2, 4, False, 10.0), (7, 2, 4, True, 10.0), (7, 3, 2, False, None),
                                                        (5, 2, 4, False, 10.0), (3, 2, 2, False, 5.0), (15, 2, 2, False, 12.0),
                                                        (15, 3, 2, True, None), (15, 2, 2, False, None)])
def test_track_default_vs_plain_kernel(ctx, ko, cfg1, img0, img1, window, levels, ss, retain, mr):
    """KLT_OPT_TRACK_VARIANT=4 (the default kernel selection) gives the same records as the plain one-feature-per-wavefront
    kernel (=0) and the oracle, and for the default context as the reference's goldens."""
    tc = make_tc(levels=levels, ss=ss, window=window, max_residue=mr, retainTrackers=retain)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, 100)
    try:
        ctx.set_option(11, 4)
        out, _ = ctx.track(0, 1, fl)
        ctx.set_option(11, 0)
        ref, _ = ctx.track(0, 1, fl)
    finally:
        ctx.set_option(11, 4)               # the default
    assert np.array_equal(out, ref), "tracker variants disagree"
    a0, a1 = np.asarray(img0, np.float32), np.asarray(img1, np.float32)
    ofl = ko.select_good_features(p, a0, 100)
    ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), ofl)
    assert_feats(out, *oracle_feats(ofl), what="tracker, window %d" % window)
    if (window, levels, ss, retain, mr) == (7, 2, 4, False, 10.0):
        assert_feats(out, cfg1["trk100_r10_x"], cfg1["trk100_r10_y"], cfg1["trk100_r10_val"], "tracker vs golden")


@pytest.mark.parametrize("mr,retain", [(10.0, False), (None, False), (10.0, True)])
def test_track_quad_kernel_long_list(ctx, ko, mr, retain):
    """The default tracker for 7x7 windows and lists of 2048 features and more (four features per wavefront, 16-byte loads) against
    the one-feature-per-wavefront kernel and the oracle, with lost features and a list length that is not a multiple of four."""
    from pyfeaturetrack_amd import synth
    f0, f1 = synth.synth_pair(1280, 720, 6)
    tc = make_tc(levels=3, ss=4, max_residue=mr, retainTrackers=retain)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, placed = ctx.select(0, 2303)
    assert placed == 2303
    fl["val"][5::11] = -4
    try:
        ctx.set_option(11, 4)
        out, _ = ctx.track(0, 1, fl)
        ctx.set_option(11, 0)
        ref, _ = ctx.track(0, 1, fl)
    finally:
        ctx.set_option(11, 4)
    assert np.array_equal(out, ref)
    a0, a1 = f0.astype(np.float32), f1.astype(np.float32)
    ofl = ko.select_good_features(p, a0, 2303)
    ofl["val"][5::11] = -4
    ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), ofl)
    assert_feats(out, *oracle_feats(ofl), what="quad tracker, 2303 features")


def test_track_xcd_aware_order(ctx, ko):
    """KLT_OPT_TRACK_XCD_ORDER (default on): features handed to the tracker sorted by row, one band per XCD, the order kept for
    later launches of the same length -- same records as in list order, lost features (passed through) included."""
    from pyfeaturetrack_amd import synth
    f0, f1 = synth.synth_pair(1280, 720, 4)
    tc = make_tc(levels=3, ss=4, max_residue=10.0)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, 1500)
    fl["val"][::7] = -3                     # some lost features in the list
    ref = None
    try:
        for variant in (0, 4):
            ctx.set_option(11, variant)
            ctx.set_option(13, 0)
            ref, _ = ctx.track(0, 1, fl)
            ctx.set_option(13, 1)
            out, _ = ctx.track(0, 1, fl)                 # sorts
            out2, _ = ctx.track(0, 1, fl[::-1].copy())   # same length, other list: the stored order is reused
            assert np.array_equal(out, ref), variant
            assert np.array_equal(out2[::-1], ref), variant
    finally:
        ctx.set_option(13, 1)
        ctx.set_option(11, 4)
    assert np.array_equal(out["val"][::7], fl["val"][::7])


def test_pingpong_and_lost_features_skipped(ctx, cfg1, img0, img1):
    ctx.configure(make_tc(max_residue=10.0))
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, 50)
    for k in range(6):
        fl, _ = ctx.track(k % 2, (k + 1) % 2, fl)
        assert_feats(fl, cfg1["pp50_%d_x" % k], cfg1["pp50_%d_y" % k], cfg1["pp50_%d_val" % k], "ping-pong call %d" % k)


def test_sequential_swap_slots(ctx, cfg1, img0, img1):
    ctx.configure(make_tc(max_residue=10.0))
    ctx.upload(0, img0)
    ctx.build_pyramids(0)
    fl, _ = ctx.select(0, 50)
    ctx.upload(1, img1)
    ctx.build_pyramids(1)
    fl, _ = ctx.track(0, 1, fl)
    assert_feats(fl, cfg1["seq50_0_x"], cfg1["seq50_0_y"], cfg1["seq50_0_val"], "sequential call 0")
    ctx.swap_slots(0, 1)                           # frame-2 pyramids become frame 1
    ctx.upload(1, img0)
    ctx.build_pyramids(1)
    fl, _ = ctx.track(0, 1, fl)
    assert_feats(fl, cfg1["seq50_1_x"], cfg1["seq50_1_y"], cfg1["seq50_1_val"], "sequential call 1")


def test_synth251_select_track(ctx, synth251):
    fr = synth251_frames()
    tc = make_tc(levels=3, ss=2, max_residue=10.0)
    ctx.configure(tc)
    for s in range(3):
        ctx.upload(s, fr[s])
        ctx.build_pyramids(s)
    ctx.upload(3, fr[0])
    fl, _ = ctx.select(3, 60)
    assert_same(ctx.select_intermediate(3), synth251["sel_val"], "synth eigenvalue map")
    assert_feats(fl, synth251["sel60_x"], synth251["sel60_y"], synth251["sel60_val"], "synth select 60")
    fl2, _ = ctx.select(0, 60, use_pyramid=True)   # level 0 of the pyramids is the same smoothed image
    assert_feats(fl2, synth251["sel60_x"], synth251["sel60_y"], synth251["sel60_val"], "synth select via pyramid level 0")
    fl, _ = ctx.track(0, 1, fl)
    assert_feats(fl, synth251["trk_0_x"], synth251["trk_0_y"], synth251["trk_0_val"], "synth track 0->1")
    fl, _ = ctx.track(1, 2, fl)
    assert_feats(fl, synth251["trk_1_x"], synth251["trk_1_y"], synth251["trk_1_val"], "synth track 1->2")


def test_synth251_window15(ctx, synth251):
    fr = synth251_frames()
    tc = make_tc(levels=2, ss=2, window=15)
    ctx.configure(tc)
    ctx.upload(0, fr[0])
    ctx.upload(1, fr[1])
    fl, _ = ctx.select(0, 25)
    assert_feats(fl, synth251["w15_sel_x"], synth251["w15_sel_y"], synth251["w15_sel_val"], "15x15 select")
    tc.max_residue = 12.0
    ctx.configure(tc)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.track(0, 1, fl)
    assert_feats(fl, synth251["w15_trk_x"], synth251["w15_trk_y"], synth251["w15_trk_val"], "15x15 track")


@pytest.mark.parametrize("window,levels,ss", [(3, 2, 2), (5, 2, 4), (9, 3, 2), (11, 2, 2), (21, 2, 2), (31, 1, 2)])
def test_other_windows_vs_oracle(ctx, ko, window, levels, ss):
    from pyfeaturetrack_amd import synth
    W, H = 400, 300
    base = synth.synth_base(W, H, 5)
    f0 = synth.shift_frame(base, 0, 0)
    f1 = synth.shift_frame(base, 1.7, -1.2)
    tc = make_tc(levels=levels, ss=ss, window=window, max_residue=15.0)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, 80)
    ofl = ko.select_good_features(p, f0.astype(np.float32), 80)
    assert_feats(fl, *oracle_feats(ofl), what="window %d select" % window)
    out, _ = ctx.track(0, 1, fl)
    ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
    assert_feats(out, *oracle_feats(ofl), what="window %d track" % window)


# ------------------------------------------------------------------------ full-size (cfg-2)
def test_cfg2_1080p_vs_oracle_and_known_shift(ctx, ko):
    """BASELINE cfg-2: 1920x1080, 5000 features, 7x7, 3 levels / ss 4 (border 120)."""
    from pyfeaturetrack_amd import synth
    f0, f1 = synth.synth_pair(1920, 1080, 1)
    tc = make_tc(levels=3, ss=4)
    assert tc.borderx == 120.0
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids(0, sync=False)
    ctx.build_pyramids(1, sync=False)
    fl, placed = ctx.select(0, 5000, use_pyramid=True)
    ofl, oval = ko.select_good_features(p, f0.astype(np.float32), 5000, want_val=True)
    assert_same(ctx.select_intermediate(3), oval, "1080p eigenvalue map")
    assert placed == int(np.count_nonzero(ofl["val"] >= 0)) == 5000
    assert_feats(fl, *oracle_feats(ofl), what="1080p select 5000")
    out, k = ctx.track(0, 1, fl)
    P0, P1 = ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32))
    for l in range(3):
        for pi, w in enumerate(("img", "gx", "gy")):
            assert_same(ctx.download_level(1, pi, l), P1.level(w, l), "1080p frame-1 %s level %d" % (w, l))
    ko.track_features(p, P0, P1, ofl)
    assert_feats(out, *oracle_feats(ofl), what="1080p track")
    live = out["val"] == 0
    assert live.sum() > 4500
    dx = np.median(out["x"][live] - fl["x"][live])
    dy = np.median(out["y"][live] - fl["y"][live])
    assert abs(dx - 3.3) < 0.02 and abs(dy + 2.1) < 0.02, (dx, dy)


def test_cfg3_shape_15x15_4_levels_translation(ctx, ko):
    """BASELINE cfg-3 geometry (1920x1080, 15x15 window, 4 levels / ss 2, border 108), translation part only:
    the affine consistency check has no reference implementation to compare with."""
    from pyfeaturetrack_amd import synth
    f0, f1 = synth.synth_pair(1920, 1080, 3)
    tc = make_tc(levels=4, ss=2, window=15, max_residue=10.0)
    assert tc.borderx == 108.0
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids_batch([0, 1])
    fl, placed = ctx.select(0, 5000, use_pyramid=True)
    ofl = ko.select_good_features(p, f0.astype(np.float32), 5000)
    assert_feats(fl, *oracle_feats(ofl), what="cfg-3 select")
    out, _ = ctx.track(0, 1, fl)
    ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
    assert_feats(out, *oracle_feats(ofl), what="cfg-3 track (translation)")
    assert (out["val"] == 0).sum() > 4500


def test_cfg4_shape_batched_pairs(ctx, ko):
    """BASELINE cfg-4 geometry (1280x720 pairs, 2000 features, 7x7, 3 levels / ss 4): a shard of 6 pairs goes through
    one batched pyramid build and ONE tracker launch; every pair equals the oracle."""
    from pyfeaturetrack_amd import synth
    NP, NF = 6, 2000
    tc = make_tc(levels=3, ss=4)
    p = params_from_tc(tc)
    ctx.configure(tc)
    frames = [synth.synth_pair(1280, 720, seed) for seed in range(NP)]
    for i, (a, b) in enumerate(frames):
        ctx.upload(2 * i, a)
        ctx.upload(2 * i + 1, b)
    ctx.build_pyramids_batch(list(range(2 * NP)))
    for i in range(NP):
        ctx.select_async(2 * i, 1, True, 2 * i, NF)              # feature buffer 2i <- selection on frame 0 of pair i
    ctx.track_batch_async([(2 * i, 2 * i + 1, 2 * i, 2 * i + 1) for i in range(NP)], NF)
    for i, (a, b) in enumerate(frames):
        sel = ctx.featbuf_download(2 * i, NF)
        out = ctx.featbuf_download(2 * i + 1, NF)
        ofl = ko.select_good_features(p, a.astype(np.float32), NF)
        assert_feats(sel, *oracle_feats(ofl), what="cfg-4 pair %d select" % i)
        ko.track_features(p, ko.Pyramids(p, a.astype(np.float32)), ko.Pyramids(p, b.astype(np.float32)), ofl)
        assert_feats(out, *oracle_feats(ofl), what="cfg-4 pair %d track" % i)


def test_batched_pairs_in_alternating_sub_shards(ctx):
    """A shard tracked in one launch, and the same shard in two sub-shards that alternate step after step with their pyramids rebuilt on the
    build stream (KLT_OPT_BUILD_STREAM): identical records.  The alternating launches exercise the device-side cache of descriptor tables
    (one upload per distinct batch), the feature orders kept with them, and the once-per-event waits of the batched calls."""
    from pyfeaturetrack_amd import synth
    NP, NF = 6, 1500
    tc = make_tc(levels=3, ss=4)
    ctx.configure(tc)
    frames = [synth.synth_pair(640, 480, 40 + seed) for seed in range(NP)]
    for i, (a, b) in enumerate(frames):
        ctx.upload(2 * i, a)
        ctx.upload(2 * i + 1, b)
    ctx.build_pyramids_batch(list(range(2 * NP)))
    IN, OUT, OUT2 = 100, 200, 300
    for i in range(NP):
        ctx.select_async(2 * i, 1, True, IN + i, NF)
    whole = [(2 * i, 2 * i + 1, IN + i, OUT + i) for i in range(NP)]
    ctx.track_batch_async(whole, NF)
    ref = [ctx.featbuf_download(OUT + i, NF) for i in range(NP)]
    assert sum(int((r["val"] == 0).sum()) for r in ref) > 2000
    ctx.set_option(15, 1)
    try:
        halves = [[(2 * i, 2 * i + 1, IN + i, OUT2 + i) for i in range(0, NP // 2)], [(2 * i, 2 * i + 1, IN + i, OUT2 + i) for i in range(NP // 2, NP)]]
        for step in range(4):
            for h in halves:
                ctx.build_pyramids_batch([s for t in h for s in t[:2]])
                ctx.track_batch_async(h, NF)
        for i in range(NP):
            got = ctx.featbuf_download(OUT2 + i, NF)
            assert got.tobytes() == ref[i].tobytes(), "pair %d differs between the whole shard and its sub-shards" % i
    finally:
        ctx.set_option(15, 0)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_parameter_draws_vs_oracle(seed):
    """tests/fuzz/fuzz_parity.py: frame sizes of any parity, 1-4 levels, subsampling 2 / 4 / 8, windows 3-15, minimum distance 0-24, skipped
    pixels, borders, residue limits, iteration counts and list lengths drawn at random -- selection, tracking and the replacement of the
    lost features identical to the oracle's in every record (900 draws were run when the tool was written; 2 x 25 stay in the suite)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from pyfeaturetrack_amd.backend import Context
    rng = np.random.default_rng(seed)
    c = Context(0)
    try:
        busy = 0
        for k in range(25):
            t = fz.draw(rng, 300000)
            bad = fz.run_trial(c, t)
            assert bad is None, "draw %d of seed %d: %s differs from the oracle: %r" % (k, seed, bad, t)
            busy += "tracked 0," not in t["_stat"]
        assert busy >= 15
    finally:
        c.close()


def test_random_draws_the_reference_ran(ctx, golden_dir):
    """The HIP path on the 160 + 60 random parameter draws the reference itself ran (tests/golden/random_draws.npz, random_draws_large.npz:
    frames up to 1400 x 1000, lists up to 2000 features): selected, tracked and replaced lists equal the reference's in every record."""
    from helpers import draw_equal, random_draws
    from pyfeaturetrack_amd.backend import REPLACING_SOME
    for t, tc, f0, f1, want in random_draws(golden_dir) + random_draws(golden_dir, "random_draws_large.npz"):
        ctx.configure(tc)
        ctx.upload(0, f0)
        ctx.upload(1, f1)
        ctx.build_pyramids(0)
        ctx.build_pyramids(1)
        fl, _ = ctx.select(0, t["n"])
        assert draw_equal(fl, want["sel"]), "selection differs from the reference: %r" % (t["seed"],)
        fl, _ = ctx.track(0, 1, fl)
        assert draw_equal(fl, want["trk"]), "tracking differs from the reference: %r" % (t["seed"],)
        fl, _ = ctx.select(1, t["n"], mode=REPLACING_SOME, fl=fl)
        assert draw_equal(fl, want["rep"]), "replacement differs from the reference: %r" % (t["seed"],)
        ctx.upload(2, t["frame2"])
        ctx.build_pyramids(2)
        fl, _ = ctx.track(1, 2, fl)                    # frame 1's pyramids stay in their slot, as the reference's sequential mode keeps them
        assert draw_equal(fl, want["trk2"]), "second tracking call differs from the reference: %r" % (t["seed"],)


def test_random_sequences_vs_per_frame_api():
    """tests/fuzz/fuzz_parity.py --sequence: KLTTrackSequence (device-resident table; build stream, prepared scores and frame stager switched on
    and off by the draw) against the per-frame host API loop on 20 short random sequences with a wiped region -- identical tables."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from pyfeaturetrack_amd import selectGoodFeatures as sgf, trackFeatures as tf
    verbose = sgf.KLT_verbose, tf.KLT_verbose
    rng = np.random.default_rng(77)
    try:
        replaced = 0
        for k in range(20):
            t = fz.draw(rng, 250000, 500, 700)
            bad = fz.run_sequence_trial(t)
            assert bad is None, "draw %d: %s differs: %r" % (k, bad, t)
            replaced += " replaced 0" not in t["_stat"]
        assert replaced >= 5
    finally:
        sgf.KLT_verbose, tf.KLT_verbose = verbose


def test_kernel_timing_by_event_pairs_and_by_dispatch_timestamps(ctx):
    """klt_timing_enable: 1 = an event pair around every launch, 2 = the level-0 pyramid launch by the start / stop events of its own
    dispatch (what a profiler reports as the kernel's duration).  Both see every launch; the dispatch figure is usually the smaller one (the pair
    also holds the boundary between two dependent launches) and never implausibly far from it; the other families are timed alike in both modes."""
    from pyfeaturetrack_amd import synth
    ctx.configure(make_tc(levels=3, ss=4))
    for k, f in enumerate(synth.synth_pair(1920, 1080, 1)):
        ctx.upload(k, f)
    figures = {}
    for mode in (1, 2):
        ctx.build_pyramids_batch([0, 1], sync=True)
        ctx.timing_enable(mode)
        for _ in range(20):
            ctx.build_pyramids_batch([0, 1])
        figures[mode] = {k["name"]: (k["launches"], 1e3 * k["total_ms"] / max(k["launches"], 1)) for k in ctx.timing_read()}
        ctx.timing_enable(False)
    pair, stamp = figures[1]["smooth_grad_l0"], figures[2]["smooth_grad_l0"]
    assert pair[0] == stamp[0] == 20
    assert 0.6 * pair[1] < stamp[1] < 1.1 * pair[1], (pair, stamp)     # (the two have been seen within 1 % of each other: no strict order)
    assert figures[1]["pyramid_reduce"][0] == figures[2]["pyramid_reduce"][0] > 0


def test_more_distinct_batches_than_the_table_cache_holds(ctx, img0, img1):
    """klt_track_batch_async keeps the descriptor tables of the last 256 distinct batches on the device: 300 batches that differ in their
    output buffers (then the first ones again, whose tables have been replaced meanwhile) all give the records of the first."""
    ctx.configure(make_tc(max_residue=10.0))
    for k, im in enumerate((img0, img1, img0, img1)):
        ctx.upload(k, im)
    ctx.build_pyramids_batch([0, 1, 2, 3], sync=True)
    N, NB = 100, 300
    fl, _ = ctx.select(0, N, use_pyramid=True)
    ctx.featbuf_upload(700, fl)
    ctx.featbuf_alloc(701, 2 * NB * N)
    for k in range(2 * NB):
        ctx.featbuf_view(1000 + k, 701, k * N, N)
    order = list(range(NB)) + [0, 1, 2]
    for k in order:
        ctx.track_batch_async([(0, 1, 700, 1000 + 2 * k), (2, 3, 700, 1000 + 2 * k + 1)], N)
    out = ctx.featbuf_download(701, 2 * NB * N).reshape(2 * NB, N)
    assert (out[0]["val"] == 0).sum() > N // 2
    for k in range(1, 2 * NB):
        assert out[k].tobytes() == out[0].tobytes(), "batch %d differs" % (k // 2)


def test_cfg5_shape_4k_sequence_with_replacement(ctx, ko):
    """BASELINE cfg-5 geometry (3840x2160, 20000 features, sequential mode, lost features replaced after every
    frame), three frames.  Pinned at the level the reference implements: tracking + _enforceMinimumDistance in
    REPLACING_SOME mode on the level-0 images kept from the last track (selectGoodFeatures.py:176-181)."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import REPLACING_SOME
    W, H, NF = 3840, 2160, 20000
    base = synth.synth_base(W, H, 4)
    frames = [synth.synth_frame(W, H, 4, k, base=base) for k in range(3)]
    tc = make_tc(levels=3, ss=4, max_residue=10.0)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(0, frames[0])
    ctx.build_pyramids(0)
    fl, placed = ctx.select(0, NF, use_pyramid=True)
    ofl = ko.select_good_features(p, frames[0].astype(np.float32), NF)
    assert placed == NF
    assert_feats(fl, *oracle_feats(ofl), what="cfg-5 initial select")
    P_prev = ko.Pyramids(p, frames[0].astype(np.float32))
    for k in (1, 2):
        ctx.upload(1, frames[k])
        ctx.build_pyramids(1)
        fl, _ = ctx.track(0, 1, fl)
        ctx.swap_slots(0, 1)                                       # sequential mode
        P_cur = ko.Pyramids(p, frames[k].astype(np.float32))
        ko.track_features(p, P_prev, P_cur, ofl)
        assert_feats(fl, *oracle_feats(ofl), what="cfg-5 track into frame %d" % k)
        lost = int(np.count_nonzero(fl["val"] < 0))
        fl, replaced = ctx.select(0, NF, mode=REPLACING_SOME, fl=fl, use_pyramid=True)
        ofl = ko.select_good_features(p, frames[k].astype(np.float32), NF, mode=2, fl=ofl)
        assert replaced == lost
        assert_feats(fl, *oracle_feats(ofl), what="cfg-5 replace after frame %d" % k)
        P_prev = P_cur


def _run_bench(extra_args, **env_kw):
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE")}
    env.update(env_kw)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra_args, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path -- rendezvous file, klt_comm_init_rank, the all-gather of the device-side record table on
    libkltgpu's side stream, barrier / max over ranks through klt_comm_allreduce_max -- with a single rank (no torch)."""
    line = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "5", "--resident-pairs", "8", "--inflight", "2", "--batch", "2", "--no-cpu-baseline", "--no-extras"],
                      KLT_FORCE_DIST="1", KLT_RDZV_FILE=str(tmp_path / "ids"))
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["tracked"] > 4500
    assert line["config"]["rccl_ranks"] == 1 and line["parity_checked"] is True and line["max_abs_dx"] <= 1e-3
    assert line["parity_cases"] == 8 and line["config"]["resident_pairs"] == 8


def test_cfg4_sharded_bench_with_native_gather(tmp_path):
    """`--config cfg4` on one rank with the RCCL path forced: batched build, one tracker launch, one gather of the table."""
    line = _run_bench(["--config", "cfg4", "--gpus", "1", "--pairs", "4", "--steps", "3", "--warmup", "1", "--repeats", "5"],
                      KLT_FORCE_DIST="1", KLT_RDZV_FILE=str(tmp_path / "ids"))
    assert line["config"]["pairs_per_step"] == 4 and line["config"]["rccl_ranks"] == 1 and line["config"]["gathered_table_ok"] is True
    assert line["parity_checked"] is True and line["scaling"] == "strong" and line["value"] > 0
    assert 0 < line["roofline"]["step_frac"] < 1 and line["roofline"]["kernels"]["track"]["timed_by"] == "dispatch timestamps"


def test_cfg5_blocks_with_the_baton_on_one_rank(tmp_path):
    """`bench.py --config cfg5 --gpus N`: one 4K sequence in blocks of frames, the feature list handed from rank to rank
    (klt_sendrecv_featbuf_async), every block's pyramids and selection scores enqueued ahead on its owner's build stream
    (KLT_OPT_SCORE_SETS).  With one rank the baton is a device copy through the same entry point; the sequence keeps all its
    features alive, as the un-blocked sequence does."""
    line = _run_bench(["--config", "cfg5", "--gpus", "1", "--steps", "7", "--repeats", "5"], KLT_FORCE_DIST="1",
                      KLT_RDZV_FILE=str(tmp_path / "ids"))
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["rccl_ranks"] == 1 and line["value"] > 0
    assert cfg["baton_copy_ok"] is True and cfg["live_after_each_block"] == [20000]
    plain = _run_bench(["--config", "cfg5", "--frames", "8", "--steps", "7", "--repeats", "5", "--min-timed-s", "1"])
    assert plain["config"]["live_at_end"] == 20000 and plain["config"]["scores_prepared"] is True
    assert plain["parity_checked"] is True and plain["roofline"]["step_frac"] < 1 and plain["cpu_baseline"]["value"] > 0


def test_sendrecv_featbuf_to_oneself():
    """klt_sendrecv_featbuf_async on a one-rank communicator: the device copy, the wait of the context's stream for the arrival,
    argument checks."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    c = Context(0)
    try:
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.sendrecv_featbuf(0, 0, 1, 0, 4)
        import ctypes as C
        from pyfeaturetrack_amd._abi import load_library
        uid = (C.c_uint8 * 128)()
        assert load_library().klt_comm_unique_id(uid) == 0
        c.comm_init(1, 0, bytes(uid))
        fl = np.zeros(1000, FEAT_DTYPE)
        fl["x"] = np.arange(1000)
        fl["val"] = 7
        c.featbuf_upload(0, fl)
        c.sendrecv_featbuf(0, 0, 1, 0, 1000)
        assert np.array_equal(c.featbuf_download(1, 1000), fl)
        with pytest.raises(KltBackendError, match="out of range"):
            c.sendrecv_featbuf(0, 1, -1, -1, 10)            # there is no rank 1
        with pytest.raises(KltBackendError, match="itself"):
            c.sendrecv_featbuf(0, 0, -1, -1, 10)            # a send to oneself needs its receive
    finally:
        c.close()


def test_native_gather_entry_points(tmp_path):
    """klt_comm_* / klt_(all)gather_featbuf_async through the C ABI on a one-rank communicator: the gathered table equals the
    source, the per-buffer fence orders a later overwrite behind the collective, errors come back as codes."""
    import ctypes as C
    from pyfeaturetrack_amd._abi import load_library
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    lib = load_library()
    c = Context(0)
    try:
        assert c.comm_info() == (1, 0)
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.allgather_featbuf_async(0, 1, 4)
        uid = (C.c_uint8 * 128)()
        assert lib.klt_comm_unique_id(uid) == 0 and any(bytes(uid))
        c.comm_init(1, 0, bytes(uid))
        with pytest.raises(KltBackendError, match="already"):
            c.comm_init(1, 0, bytes(uid))
        n = 3000
        fl = np.zeros(n, FEAT_DTYPE)
        fl["x"] = np.arange(n)
        fl["y"] = -np.arange(n)
        fl["val"] = np.arange(n) % 7 - 3
        c.featbuf_upload(0, fl)
        c.allgather_featbuf_async(0, 1, n)
        c.gather_featbuf_async(0, 2, n, root=0)
        c.comm_fence_featbuf(0)                        # the overwrite below waits for both collectives on the device
        c.featbuf_upload(0, np.zeros(n, FEAT_DTYPE))
        assert np.array_equal(c.featbuf_download(1, n), fl) and np.array_equal(c.featbuf_download(2, n), fl)
        assert c.comm_allreduce_max([1.5, -2.0]) == [1.5, -2.0]
        with pytest.raises(KltBackendError, match="root"):
            c.gather_featbuf_async(0, 2, n, root=3)
        c.comm_wait()
        c.comm_destroy()
        assert c.comm_info() == (1, 0)
    finally:
        c.close()


def test_topk_prefilter_and_its_fallback(ctx, ko):
    """At 1080p only the best candidates are sorted (KLT_OPT_TOPK_PREFILTER).  Same result as the full sort; when the
    greedy walk needs more candidates than were kept (large mindist, more features than fit) the full sort takes over."""
    from pyfeaturetrack_amd import synth
    f0 = synth.synth_frame(1920, 1080, 6, 0)
    for mindist, n in ((10, 5000), (60, 4000), (25, 1500)):
        tc = make_tc(levels=3, ss=4, mindist=mindist)
        ctx.configure(tc)
        ctx.upload(0, f0)
        ofl = ko.select_good_features(params_from_tc(tc), f0.astype(np.float32), n)
        fl, placed = ctx.select(0, n)
        assert placed == int(np.count_nonzero(ofl["val"] >= 0))
        assert_feats(fl, *oracle_feats(ofl), what="prefiltered select mindist=%d n=%d" % (mindist, n))
        try:
            ctx.set_option(5, 0)
            fl2, _ = ctx.select(0, n)
        finally:
            ctx.set_option(5, 1)
        assert_feats(fl2, *oracle_feats(ofl), what="full-sort select mindist=%d n=%d" % (mindist, n))


def test_replacement_cut_and_its_repeat(ctx, ko):
    """REPLACING_SOME at 1080p keeps 64 candidates per LOST feature (at least 4096).  Few lost features: the cut holds.  Many lost
    features under a large minimum distance: the kept candidates run out and the selection repeats with every candidate.  Either way
    the list equals the reference walk's (selectGoodFeatures.py:45-135, :279-294)."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import REPLACING_SOME
    f0 = synth.synth_frame(1920, 1080, 6, 0)
    for mindist, n, lose in ((10, 5000, 40), (40, 700, 450), (25, 1500, 1500)):
        tc = make_tc(levels=3, ss=4, mindist=mindist)
        ctx.configure(tc)
        ctx.upload(0, f0)
        p = params_from_tc(tc)
        base = ko.select_good_features(p, f0.astype(np.float32), n)
        live = np.flatnonzero(base["val"] >= 0)
        rng = np.random.default_rng(mindist)
        gone = rng.choice(live, min(lose, len(live)), replace=False)
        fl = base.copy()
        fl["val"][gone] = -1
        fl["x"][gone] = -1.0
        fl["y"][gone] = -1.0
        want = ko.select_good_features(p, f0.astype(np.float32), n, mode=REPLACING_SOME, fl=fl.copy())
        got, replaced = ctx.select(0, n, mode=REPLACING_SOME, fl=fl.copy())
        assert replaced == int(np.count_nonzero((want["val"] >= 0) & (fl["val"] < 0)))
        assert_feats(got, *oracle_feats(want), what="replace mindist=%d lost=%d of %d" % (mindist, len(gone), n))


@pytest.mark.parametrize("build_stream", [0, 1])
def test_select_prepare_gives_the_same_replacement(ko, build_stream):
    """klt_select_prepare_async scores a slot ahead of its replacement pass (on the build stream when that is on).  Same list as the
    inline path and as the reference walk; scores that no longer belong to the slot's contents or to the parameters are not used."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context, REPLACING_SOME
    n = 3000
    base = synth.synth_base(1920, 1080, 7)
    f0, f1 = synth.synth_frame(1920, 1080, 7, 0, base=base), synth.synth_frame(1920, 1080, 7, 3, base=base)
    tc = make_tc(levels=3, ss=4)
    p = params_from_tc(tc)
    c = Context(0)
    try:
        c.configure(tc)
        c.set_option(15, build_stream)

        def lose(fl, seed):
            rng = np.random.default_rng(seed)
            gone = rng.choice(np.flatnonzero(fl["val"] >= 0), 120, replace=False)
            fl = fl.copy()
            fl["val"][gone] = -1
            fl["x"][gone] = -1.0
            fl["y"][gone] = -1.0
            return fl

        def check(slot, frame, fl, what):
            want = ko.select_good_features(p, frame.astype(np.float32), n, mode=REPLACING_SOME, fl=fl.copy())
            got, _ = c.select(slot, n, mode=REPLACING_SOME, fl=fl.copy(), use_pyramid=True)
            assert_feats(got, *oracle_feats(want), what=what)
            return got

        c.upload(0, f0)
        c.build_pyramids(0, sync=False)
        first, placed = c.select(0, n, use_pyramid=True)
        assert placed == n
        fl = lose(first, 1)
        inline = check(0, f0, fl, "inline replacement")
        c.select_prepare(0)
        prepared = check(0, f0, fl, "prepared replacement")
        assert np.array_equal(inline, prepared)
        from pyfeaturetrack_amd.backend import KltBackendError
        with pytest.raises(KltBackendError, match="prepared scores"):
            c.select_intermediate(3)                    # that selection wrote no eigenvalue map
        # the scores follow the slot's contents through a swap ...
        c.upload(1, f1)
        c.build_pyramids(1, sync=False)
        c.select_prepare(1)
        c.swap_slots(0, 1)
        check(0, f1, fl, "prepared, then swapped")
        # ... and are dropped when the slot is rebuilt with another frame
        c.select_prepare(0)
        c.upload(0, f0)
        c.build_pyramids(0, sync=False)
        check(0, f0, lose(first, 2), "prepared, then rebuilt")
        # ... or when the selection parameters change
        c.select_prepare(0)
        tc2 = make_tc(levels=3, ss=4, mindist=14)
        tc2.min_eigenvalue = 40
        c.configure(tc2)
        p = params_from_tc(tc2)
        c.build_pyramids(0, sync=False)                 # (set_params keeps the pyramids: same taps)
        check(0, f0, lose(first, 3), "prepared, then other parameters")
    finally:
        c.close()


def test_replacement_seed_map_stamps_wrap(ctx, cfg1, img1, ko):
    """The map of the live features' squares is stamped per replacement pass and cleared only when the 255 stamps wrap (or the frame
    size changes): more than 255 passes in one context, other frame sizes in between, still the reference walk's list."""
    from pyfeaturetrack_amd.backend import REPLACING_SOME
    tc = make_tc(max_residue=10.0)
    p = params_from_tc(tc)
    ctx.configure(tc)
    ctx.upload(2, img1)
    small = np.ascontiguousarray(img1[:200, :300])
    ctx.upload(3, small)
    base = ko.select_good_features(p, img1.astype(np.float32), 100)
    rng = np.random.default_rng(9)
    for it in range(300):
        fl = base.copy()
        gone = rng.choice(100, 1 + it % 37, replace=False)
        fl["val"][gone] = -1
        fl["x"][gone] = -1.0
        fl["y"][gone] = -1.0
        frame, slot = (small, 3) if it % 50 == 49 else (img1, 2)
        if slot == 3:
            keep = (fl["x"] < 290) & (fl["y"] < 190)
            fl["val"][~keep] = -1
        got, _ = ctx.select(slot, 100, mode=REPLACING_SOME, fl=fl.copy())
        if it % 50 == 49 or it >= 250 or it < 3:
            want = ko.select_good_features(p, frame.astype(np.float32), 100, mode=REPLACING_SOME, fl=fl.copy())
            assert_feats(got, *oracle_feats(want), what="replacement pass %d" % it)


def test_8k_frame_select_replace_and_track(ko):
    """Twice the width and height of the largest BASELINE frame (7680x4320, 33 Mpixel; 32 400 minimum-distance tiles, candidate
    indices past 2^24): selection, tracking into a shifted frame and replacement equal the oracle -- no index arithmetic overflows."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context, REPLACING_SOME
    W, H, NF = 7680, 4320, 30000
    base = synth.synth_base(W, H, 11)
    f0, f1 = synth.synth_frame(W, H, 11, 0, base=base), synth.synth_frame(W, H, 11, 1, base=base)
    tc = make_tc(levels=3, ss=4, max_residue=10.0)
    p = params_from_tc(tc)
    c = Context(0)
    try:
        c.configure(tc)
        c.upload(0, f0)
        c.upload(1, f1)
        c.build_pyramids_batch([0, 1], sync=False)
        fl, placed = c.select(0, NF, use_pyramid=True)
        ofl = ko.select_good_features(p, f0.astype(np.float32), NF)
        assert placed == NF
        assert_feats(fl, *oracle_feats(ofl), what="8K select")
        fl, _ = c.track(0, 1, fl)
        ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
        assert_feats(fl, *oracle_feats(ofl), what="8K track")
        assert int(np.count_nonzero(fl["val"] >= 0)) > NF * 0.9
        c.select_prepare(1)
        fl, _ = c.select(1, NF, mode=REPLACING_SOME, fl=fl, use_pyramid=True)
        ofl = ko.select_good_features(p, f1.astype(np.float32), NF, mode=REPLACING_SOME, fl=ofl)
        assert_feats(fl, *oracle_feats(ofl), what="8K replacement (prepared scores)")
    finally:
        c.close()


def test_select_begin_finish(ko):
    """klt_select_begin_async / klt_select_finish: the selection in two halves with other work (upload, build and score preparation of
    another slot) enqueued in between; the same list as klt_select_async; a second selection while one is pending is refused."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context, KltBackendError, REPLACING_SOME, SELECTING_ALL
    n = 3000
    base = synth.synth_base(1920, 1080, 13)
    f0, f1 = synth.synth_frame(1920, 1080, 13, 0, base=base), synth.synth_frame(1920, 1080, 13, 2, base=base)
    tc = make_tc(levels=3, ss=4)
    p = params_from_tc(tc)
    c = Context(0)
    try:
        c.configure(tc)
        c.set_option(15, 1)
        c.select_finish()                                   # nothing pending: a no-op
        c.upload(0, f0)
        c.build_pyramids(0, sync=False)
        c.featbuf_alloc(0, n)
        c.select_begin(0, SELECTING_ALL, True, 0, n)
        with pytest.raises(KltBackendError, match="pending"):
            c.select_begin(0, SELECTING_ALL, True, 0, n)
        c.upload(1, f1)                                     # other work between the halves
        c.build_pyramids(1, sync=False)
        c.select_prepare(1)
        c.select_finish()
        first = c.featbuf_download(0, n)
        want = ko.select_good_features(p, f0.astype(np.float32), n)
        assert_feats(first, *oracle_feats(want), what="selection in two halves")
        fl = first.copy()
        gone = np.random.default_rng(3).choice(n, 200, replace=False)
        fl["val"][gone] = -1
        fl["x"][gone] = -1.0
        fl["y"][gone] = -1.0
        c.featbuf_upload(1, fl)
        c.select_begin(1, REPLACING_SOME, True, 1, n)       # uses the scores prepared above
        c.build_pyramids(0, sync=False)                     # the other slot is rebuilt and re-scored meanwhile
        c.select_prepare(0)
        c.select_finish()
        got = c.featbuf_download(1, n)
        want = ko.select_good_features(p, f1.astype(np.float32), n, mode=REPLACING_SOME, fl=fl.copy())
        assert_feats(got, *oracle_feats(want), what="replacement in two halves")
    finally:
        c.close()


def test_nms_global_grid_path(ctx, ko):
    """mindist 2 at 1920x1080 -> the cell grid (960x540 u32) exceeds LDS and lives in global memory"""
    from pyfeaturetrack_amd import synth
    f0 = synth.synth_frame(1920, 1080, 2, 0)
    tc = make_tc(levels=3, ss=4, mindist=2)
    ctx.configure(tc)
    ctx.upload(0, f0)
    fl, _ = ctx.select(0, 4000)
    ofl = ko.select_good_features(params_from_tc(tc), f0.astype(np.float32), 4000)
    assert_feats(fl, *oracle_feats(ofl), what="select mindist=2 (global grid)")


def test_abi_error_reporting(ctx, img0):
    """Errors come back as negative status codes with a message; nothing exits or throws across the ABI."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError, FEAT_DTYPE
    c = Context(0)
    try:
        with pytest.raises(KltBackendError, match="klt_set_params"):
            c.upload(0, img0)
            c.build_pyramids(0)                              # parameters never set
        c.configure(make_tc())
        c.upload(0, img0)
        c.upload(1, img0[:100, :200].copy())
        c.build_pyramids_batch([0, 1], sync=True)            # different sizes: two launch groups, fine
        fl = np.zeros(10, FEAT_DTYPE)
        with pytest.raises(KltBackendError, match="differ in size"):
            c.track(0, 1, fl)
        with pytest.raises(KltBackendError, match="no frame"):
            c.build_pyramids(7)
        c.upload(2, img0)
        with pytest.raises(KltBackendError, match="pyramids"):
            c.track(0, 2, fl)                                # slot 2 has a frame but no pyramids
        p = params_from_tc(make_tc())
        p.window_width = p.window_height = 8
        with pytest.raises(KltBackendError, match="window"):
            c.set_params(p)
        with pytest.raises(KltBackendError, match="image too small"):
            c.configure(make_tc(levels=4, ss=8))
            c.upload(3, img0[:40, :40].copy())
            c.build_pyramids(3)
    finally:
        c.close()


def test_async_ingest_from_pinned_memory(cfg1, img0, img1):
    """klt_upload_u8_async: frames copied from pinned memory on the copy stream; builds wait for the copy, copies wait
    for the builds that still read the slot; results equal the synchronous path."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(max_residue=10.0))
        p0, p1, scratch = c.pinned_array(img0.shape), c.pinned_array(img1.shape), c.pinned_array(img1.shape)
        p0[:] = img0
        p1[:] = img1
        scratch[:] = 255 - img1
        c.upload_async(0, p0)
        c.upload_async(1, p1)
        c.build_pyramids_batch([0, 1])
        fl, _ = c.select(0, 100, use_pyramid=True)
        c.featbuf_upload(0, fl)
        for i in range(12):
            c.upload_async(1, scratch if i % 2 == 0 else p1)      # overwrite frame 2 while earlier work may be queued
            c.upload_async(1, p1)
            c.build_pyramids_batch([1])
            c.track_async(0, 1, 0, 1 + i % 2, 100)
        for k in (1, 2):
            out = c.featbuf_download(k, 100)
            assert_feats(out, cfg1["trk100_r10_x"], cfg1["trk100_r10_y"], cfg1["trk100_r10_val"], "async ingest, buffer %d" % k)
        with pytest.raises(KltBackendError, match="pinned"):
            c.upload_async(1, np.ascontiguousarray(img1))         # pageable memory is refused
    finally:
        c.close()


def test_feature_table_views(ctx, cfg1, img0, img1):
    """klt_featbuf_view: tracker output written into a window of a larger device-side record table"""
    ctx.configure(make_tc(max_residue=10.0))
    ctx.upload(0, img0)
    ctx.upload(1, img1)
    ctx.build_pyramids_batch([0, 1])
    fl, _ = ctx.select(0, 100, use_pyramid=True)
    ctx.featbuf_upload(40, fl)
    ctx.featbuf_alloc(41, 300)
    ctx.featbuf_view(42, 41, 100, 100)
    ctx.track_async(0, 1, 40, 42, 100)
    table = ctx.featbuf_download(41, 300)
    assert np.all(table["val"][:100] == -1) and np.all(table["val"][200:] == -1)
    assert_feats(table[100:200], cfg1["trk100_r10_x"], cfg1["trk100_r10_y"], cfg1["trk100_r10_val"], "track into a table view")


# ------------------------------------------------------------------------------- Python API
def test_python_api_example1_flow(cfg1, golden_dir, tmp_path, capsys, monkeypatch):
    """The reference's example1.py call sequence through the reference-shaped API (PIL images)."""
    PIL = pytest.importorskip("PIL.Image")
    from pyfeaturetrack_amd import selectGoodFeatures as _sgf, trackFeatures as _trk, writeFeatures as _wf
    for mod in (_sgf, _trk, _wf):                  # the reference's default (selectGoodFeatures.py:14); other tests switch the prints off
        if hasattr(mod, "KLT_verbose"):
            monkeypatch.setattr(mod, "KLT_verbose", 1)
    from pyfeaturetrack_amd.klt import KLT_TrackingContext, KLTCountRemainingFeatures
    from pyfeaturetrack_amd.selectGoodFeatures import KLTSelectGoodFeatures, KLTReplaceLostFeatures
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    from pyfeaturetrack_amd.writeFeatures import KLTWriteFeatureListToPPM
    tc = KLT_TrackingContext()
    tc.max_residue = 10.0
    i0 = PIL.open(os.path.join(golden_dir, "img0.pgm"))
    i1 = PIL.open(os.path.join(golden_dir, "img1.pgm"))
    fl = KLTSelectGoodFeatures(tc, i0, 50)
    assert [(f.x, f.y, f.val) for f in fl] == [(int(x), int(y), int(v)) for x, y, v in
                                               zip(cfg1["sel50_x"], cfg1["sel50_y"], cfg1["sel50_val"])]
    assert isinstance(fl[0].x, int)
    KLTWriteFeatureListToPPM(fl, i0, str(tmp_path / "feat1.ppm"))
    for k in range(4):
        KLTTrackFeatures(tc, i0 if k % 2 == 0 else i1, i1 if k % 2 == 0 else i0, fl)
        assert [f.val for f in fl] == [int(v) for v in cfg1["pp50_%d_val" % k]]
        assert np.array_equal(np.array([f.x for f in fl], np.float64), cfg1["pp50_%d_x" % k])
        assert np.array_equal(np.array([f.y for f in fl], np.float64), cfg1["pp50_%d_y" % k])
    out = capsys.readouterr().out
    assert "(KLT) Selecting the 50 best features from a 320 by 240 image...  " in out
    assert "\t45 features successfully tracked." in out
    before = KLTCountRemainingFeatures(fl)
    KLTReplaceLostFeatures(tc, i0, fl)
    assert KLTCountRemainingFeatures(fl) == 50 > before


def test_python_api_on_random_draws_the_reference_ran(golden_dir):
    """Every fourth random draw of tests/golden/random_draws.npz through the reference-shaped Python API (PIL images, a fresh
    KLT_TrackingContext per draw set up the way the generator set up the reference's): the feature objects hold what the reference's held."""
    PIL = pytest.importorskip("PIL.Image")
    from helpers import random_draws
    from pyfeaturetrack_amd import selectGoodFeatures as sgf, trackFeatures as tf
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    verbose = sgf.KLT_verbose, tf.KLT_verbose
    sgf.KLT_verbose = tf.KLT_verbose = 0
    try:
        for t, _, f0, f1, want in random_draws(golden_dir)[::4]:
            tc = KLT_TrackingContext()
            tc.window_width = tc.window_height = t["window"]
            tc.nPyramidLevels, tc.subsampling = t["levels"], t["ss"]
            tc.KLTUpdateTCBorder()
            tc.mindist, tc.nSkippedPixels, tc.smoothBeforeSelecting = t["mindist"], t["skip"], t["smooth"]
            tc.max_residue, tc.min_eigenvalue, tc.max_iterations = t["mr"], t["min_eig"], t["max_iter"]
            i0, i1 = PIL.fromarray(f0, "L"), PIL.fromarray(f1, "L")
            tc.sequentialMode = t["sequential"]
            i2 = PIL.fromarray(t["frame2"], "L")
            fl = sgf.KLTSelectGoodFeatures(tc, i0, t["n"])
            for stage in ("sel", "trk", "trk2"):
                if stage == "trk":
                    tf.KLTTrackFeatures(tc, i0, i1, fl)
                if stage == "trk2":
                    # the reference's list went through _enforceMinimumDistance in between: take its state, then make the second call
                    # (sequential draws: the first image argument is ignored, the resident pyramids of frame 1 are used)
                    rx, ry, rv = want["rep"]
                    for f, a, b, c in zip(fl, rx, ry, rv):
                        f.x, f.y, f.val = float(a), float(b), int(c)
                    tf.KLTTrackFeatures(tc, i1, i2, fl)
                x, y, v = want[stage]
                assert [int(f.val) for f in fl] == [int(a) for a in v], "%s: val differs: seed %r" % (stage, t["seed"])
                assert np.array_equal(np.array([f.x for f in fl], np.float32), x) and np.array_equal(np.array([f.y for f in fl], np.float32), y), \
                    "%s: positions differ: seed %r" % (stage, t["seed"])
    finally:
        sgf.KLT_verbose, tf.KLT_verbose = verbose


def test_python_api_sequential_mode(cfg1, golden_dir):
    PIL = pytest.importorskip("PIL.Image")
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    sgf.KLT_verbose = 0
    try:
        tc = KLT_TrackingContext()
        tc.max_residue = 10.0
        tc.sequentialMode = True
        i0 = PIL.open(os.path.join(golden_dir, "img0.pgm"))
        i1 = PIL.open(os.path.join(golden_dir, "img1.pgm"))
        fl = sgf.KLTSelectGoodFeatures(tc, i0, 50)
        KLTTrackFeatures(tc, i0, i1, fl)
        assert tc.pyramid_last is not None
        assert np.array_equal(np.array([f.x for f in fl], np.float64), cfg1["seq50_0_x"])
        KLTTrackFeatures(tc, i0, i0, fl)           # first image ignored: frame-2 pyramids of the last call are used
        assert np.array_equal(np.array([f.x for f in fl], np.float64), cfg1["seq50_1_x"])
        assert [f.val for f in fl] == [int(v) for v in cfg1["seq50_1_val"]]
    finally:
        sgf.KLT_verbose = 1


def _host_api_sequence(tc, frames, n, replace):
    """The per-frame loop KLTTrackSequence replaces (upstream example3's order: track, replace, store)."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import storeFeatures as sf
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    ft = sf.KLTCreateFeatureTable(len(frames), n)
    fl = sgf.KLTSelectGoodFeatures(tc, frames[0], n)
    sf.KLTStoreFeatureList(fl, ft, 0)
    for k in range(1, len(frames)):
        KLTTrackFeatures(tc, frames[k - 1], frames[k], fl)
        if replace:
            sgf.KLTReplaceLostFeatures(tc, frames[k], fl)
        sf.KLTStoreFeatureList(fl, ft, k)
    return ft


@pytest.mark.gpu
@pytest.mark.parametrize("replace,affine,ingest", [(True, False, True), (False, False, False), (True, True, True)])
def test_track_sequence_matches_per_frame_api(replace, affine, ingest):
    """KLTTrackSequence (device-resident [frames x features] table, one download) gives exactly the rows the per-frame
    host API loop produces; with a generator as the frame source."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n, nf = 400, 300, 300, 7
    base = synth.synth_base(w, h, 11)
    frames = [synth.synth_frame(w, h, 11, k, shift=(2.3, -1.4), base=base) for k in range(nf)]
    frames[4] = frames[4].copy()
    frames[4][90:170, 120:260] = 128                      # wipe a region: features there are lost and replaced elsewhere

    def make():
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        tc.max_residue = 10.0
        if affine:
            tc.affineConsistencyCheck = 2
        return tc

    sgf.KLT_verbose = 0
    try:
        want = _host_api_sequence(make(), frames, n, replace)
        tc = make()
        got = KLTTrackSequence(tc, (f for f in frames), n, replace_lost=replace, async_ingest=ingest)
        assert got.nFrames == nf and got.nFeatures == n
        assert (want.val[1:] < 0).sum() > 0 or replace      # the wiped region does lose features
        if replace:
            assert (got.val[4:] > 0).sum() > 0              # ... and they are replaced (new features carry their eigenvalue)
        assert np.array_equal(got.val, want.val)
        assert np.array_equal(got.x, want.x) and np.array_equal(got.y, want.y)
        # the context is left as the per-frame API leaves it: the next KLTTrackFeatures call continues the sequence
        from pyfeaturetrack_amd import storeFeatures as sf
        from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
        nxt = synth.synth_frame(w, h, 11, nf, shift=(2.3, -1.4), base=base)
        fl = sf.KLTCreateFeatureList(n)
        sf.KLTExtractFeatureList(fl, got, nf - 1)
        if not affine:
            KLTTrackFeatures(tc, frames[-1], nxt, fl)
            live = [(f.x, f.y) for f in fl if f.val == 0]
            assert len(live) > n // 2
    finally:
        sgf.KLT_verbose = 1


@pytest.mark.gpu
@pytest.mark.parametrize("kw,n", [(dict(), 100), (dict(mindist=1), 300), (dict(mindist=0), 300), (dict(mindist=3), 2000),
                                  (dict(mindist=25), 400), (dict(mindist=70), 30), (dict(nSkippedPixels=1, mindist=12), 150),
                                  (dict(nSkippedPixels=3, mindist=2), 500), (dict(min_eigenvalue=2000), 400)])
def test_parallel_min_distance_equals_serial_walk(ctx, ko, img0, kw, n):
    """KLT_OPT_SELECT_PARALLEL_NMS: both formulations of _enforceMinimumDistance (selectGoodFeatures.py:45-135) give the
    oracle's list -- exclusion radii from none to larger than the LDS tile allows (falls back to the walk), skipped
    pixels, lists that cannot be filled."""
    tc = make_tc(**kw)
    p = params_from_tc(tc)
    ctx.configure(tc)
    want = ko.select_good_features(p, img0.astype(np.float32), n)
    got = {}
    try:
        for algo in (1, 0):
            ctx.set_option(8, algo)
            ctx.upload(2, img0)
            got[algo], placed = ctx.select(2, n)
            assert placed == int((want["val"] >= 0).sum())
            assert_feats(got[algo], want["x"], want["y"], want["val"], "algo %d %r" % (algo, kw))
    finally:
        ctx.set_option(8, 1)


@pytest.mark.gpu
def test_parallel_min_distance_long_dependency_chain(ctx, ko):
    """A smooth ramp of corner strength gives a dependency chain far longer than the eight passes enqueued at first:
    the host keeps adding passes until nothing is undecided; result = the oracle's walk."""
    h, w = 200, 1200
    rng = np.random.default_rng(5)
    ys, xs = np.mgrid[0:h, 0:w]
    amp = 20.0 + 200.0 * xs / w                                   # texture contrast grows monotonically to the right
    img = np.clip(128 + amp * 0.5 * (np.sin(xs * 0.9) * np.sin(ys * 0.9)) + rng.normal(0, 0.3, (h, w)), 0, 255).astype(np.uint8)
    tc = make_tc(mindist=10)
    p = params_from_tc(tc)
    ctx.configure(tc)
    n = 1500
    want = ko.select_good_features(p, img.astype(np.float32), n)
    ctx.upload(2, img)
    got, placed = ctx.select(2, n)
    assert placed == int((want["val"] >= 0).sum())
    assert_feats(got, want["x"], want["y"], want["val"], "ramp")


def _oracle_min_distance(ko, val, ncols, nrows, tc, n, fl=None):
    p = params_from_tc(tc)
    bx, by, _, _ = ko.scan_borders(p)
    cand = ko.sorted_candidates(val, ncols, nrows, bx, by, p.nSkippedPixels)
    out = ko.make_featurelist(n) if fl is None else fl.copy()
    ko.enforce_min_distance(cand, out, ncols, nrows, p.mindist, p.min_eigenvalue, fl is None)
    return out


@pytest.mark.parametrize("case", ["ties", "plateau", "ramp", "checker", "random_replace"])
def test_min_distance_on_given_scores(ctx, ko, case):
    """_enforceMinimumDistance (selectGoodFeatures.py:45-135) driven with eigenvalue maps real frames rarely produce
    (klt_set_score_override): equal scores inside one exclusion square (rank = x, then y), a constant plateau, a
    monotone ramp (one dependency chain across the frame), a checkerboard of two values, random scores with live
    features to keep (REPLACING_SOME).  Both formulations must give the oracle's list."""
    from pyfeaturetrack_amd import synth
    ncols, nrows, n = 500, 300, 400
    tc = make_tc(mindist=10)
    ctx.configure(tc)
    frame = synth.synth_frame(ncols, nrows, 3, 0)
    p = params_from_tc(tc)
    bx, by, _, _ = ko.scan_borders(p)
    nx, ny = ncols - 2 * bx, nrows - 2 * by
    rng = np.random.default_rng(17)
    ys, xs = np.mgrid[0:ny, 0:nx]
    if case == "ties":
        val = rng.integers(1, 40, (ny, nx)).astype(np.float32) * 8.0          # only 39 distinct scores
    elif case == "plateau":
        val = np.full((ny, nx), 50.0, np.float32)
    elif case == "ramp":
        val = (10.0 + xs + 0.001 * ys).astype(np.float32)
    elif case == "checker":
        val = np.where((xs // 3 + ys // 3) % 2 == 0, 100.0, 7.0).astype(np.float32)
    else:
        val = (rng.random((ny, nx)) * 1000.0).astype(np.float32)
        val[rng.random((ny, nx)) < 0.3] = 0.5                                  # below min_eigenvalue
    fl_in = None
    if case == "random_replace":
        fl_in = ko.make_featurelist(n)
        keep = rng.choice(n, 150, replace=False)
        fl_in["x"][keep] = rng.uniform(bx, ncols - bx - 1, 150).astype(np.float32)
        fl_in["y"][keep] = rng.uniform(by, nrows - by - 1, 150).astype(np.float32)
        fl_in["val"][keep] = 0
    want = _oracle_min_distance(ko, val, ncols, nrows, tc, n, fl_in)
    ctx.upload(2, frame)
    try:
        for algo in (1, 0):
            ctx.set_option(8, algo)
            ctx.set_score_override(val)
            if fl_in is None:
                got, placed = ctx.select(2, n)
            else:
                got, placed = ctx.select(2, n, mode=2, fl=fl_in)
            assert_feats(got, *oracle_feats(want), what="%s algo %d" % (case, algo))
    finally:
        ctx.set_option(8, 1)


def test_track_sequence_without_presmoothing():
    """tc.smoothBeforeSelecting = False: KLTTrackSequence's initial selection uses the raw frame, as KLTSelectGoodFeatures does
    (selectGoodFeatures.py:183-197) -- not level 0 of the pyramid, which is always smoothed."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n = 320, 240, 120
    base = synth.synth_base(w, h, 5)
    frames = [synth.synth_frame(w, h, 5, k, shift=(1.7, 0.9), base=base) for k in range(3)]

    def make():
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        tc.smoothBeforeSelecting = False
        return tc

    sgf.KLT_verbose = 0
    try:
        want = _host_api_sequence(make(), frames, n, True)
        got = KLTTrackSequence(make(), frames, n, replace_lost=True)
        assert np.array_equal(got.val, want.val) and np.array_equal(got.x, want.x) and np.array_equal(got.y, want.y)
        smoothed = make()
        smoothed.smoothBeforeSelecting = True
        other = KLTTrackSequence(smoothed, frames, n, replace_lost=True)
        assert not np.array_equal(other.x[0], got.x[0])          # the two settings really select different features
    finally:
        sgf.KLT_verbose = 1


def test_sequential_mode_survives_parameter_changes(img0, img1, cfg1):
    """sequentialMode keeps tc.pyramid_last across calls (trackFeatures.py:152-161).  Changing a field that does not enter the
    pyramids (mindist, max_residue, min_eigenvalue) between two calls must keep the resident pyramids; a second tracking
    context with other parameters on the same device context must not break the first one either."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd.backend import default_context
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    sgf.KLT_verbose = 0
    try:
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        tc.max_residue = 10.0
        fl = sgf.KLTSelectGoodFeatures(tc, img0, 50)
        KLTTrackFeatures(tc, img0, img1, fl)
        assert [f.val for f in fl] == cfg1["seq50_0_val"].tolist()
        s1 = tc._klt_slots[0]
        assert default_context().pyramids_valid(s1)
        tc.mindist = 12                                              # not a pyramid parameter
        tc.min_eigenvalue = 2
        default_context().configure(tc)
        assert default_context().pyramids_valid(s1), "resident pyramids were invalidated by a mindist change"
        tc.mindist, tc.min_eigenvalue = 10, 1
        other = KLT_TrackingContext()                                # another context, other taps, same device context
        other.grad_sigma = 1.5
        sgf.KLTSelectGoodFeatures(other, img1, 20)
        # the taps changed under tc's resident pyramids: the next call rebuilds frame 1 from the image it is given
        # instead of failing ("pyramids of both slots must be built")
        KLTTrackFeatures(tc, img1, img0, fl)
        assert [f.val for f in fl] == cfg1["seq50_1_val"].tolist()
        assert [f.x for f in fl] == cfg1["seq50_1_x"].tolist()
    finally:
        sgf.KLT_verbose = 1


def test_slots_and_affine_states_are_recycled():
    """Tracking contexts and feature lists that die give their device slots / affine states back (weakref finalizers)."""
    import gc
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import default_context
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    ctx = default_context()
    f0, f1 = synth.synth_pair(320, 240, seed=3)
    sgf.KLT_verbose = 0
    try:
        bases, states = set(), set()
        for _ in range(6):
            tc = KLT_TrackingContext()
            tc.affineConsistencyCheck = 2
            fl = sgf.KLTSelectGoodFeatures(tc, f0, 40)
            KLTTrackFeatures(tc, f0, f1, fl)
            bases.add(tc._klt_slots[0])
            states.update(e[1] for e in ctx.__dict__.get("_affine_states", {}).values())
            del tc, fl
            gc.collect()
        assert len(bases) <= 2 and len(states) <= 2, (bases, states)
        assert not ctx.__dict__.get("_affine_states")
    finally:
        sgf.KLT_verbose = 1


def test_build_stream_prefetch_keeps_results(ko):
    """KLT_OPT_BUILD_STREAM: pyramids of frame k+1 built on a second HIP stream while frame k is tracked and replaced (ring of three
    slots, one event each way per frame) -- the same table as with everything on one stream, with and without asynchronous ingest,
    and slots rewritten while earlier work may still be queued."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n, nf = 640, 480, 800, 9
    base = synth.synth_base(w, h, 21)
    frames = [synth.synth_frame(w, h, 21, k, shift=(2.1, 1.2), base=base) for k in range(nf)]

    def make():
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        tc.max_residue = 10.0
        return tc

    sgf.KLT_verbose = 0
    try:
        ref = KLTTrackSequence(make(), frames, n, prefetch=False)
        for ingest in (True, False):
            for _ in range(2):
                got = KLTTrackSequence(make(), (f for f in frames), n, prefetch=True, async_ingest=ingest)
                assert np.array_equal(got.rec, ref.rec), "prefetching builds changed the feature table (ingest=%s)" % ingest
        assert (ref.val[-1] >= 0).sum() > n // 2
    finally:
        sgf.KLT_verbose = 1


def test_track_sequence_frame_source_errors_surface():
    """The frames are read and staged on a helper thread: an exception of the frame source, and a frame of another size, surface in the
    caller as they did when the caller read the frames itself; the next sequence on the same context is unaffected."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n = 320, 240, 100
    base = synth.synth_base(w, h, 3)
    frames = [synth.synth_frame(w, h, 3, k, base=base) for k in range(8)]

    def make():
        tc = KLT_TrackingContext()
        tc.sequentialMode = True
        return tc

    def failing():
        for k, f in enumerate(frames):
            if k == 5:
                raise ValueError("camera unplugged")
            yield f

    def wrong_size():
        for k, f in enumerate(frames):
            yield f[:200, :300].copy() if k == 4 else f

    sgf.KLT_verbose = 0
    try:
        with pytest.raises(ValueError, match="camera unplugged"):
            KLTTrackSequence(make(), failing(), n)
        with pytest.raises(SystemExit):                     # KLTError prints and exits, as the reference's does
            KLTTrackSequence(make(), wrong_size(), n)
        a = KLTTrackSequence(make(), iter(frames), n)
        b = KLTTrackSequence(make(), iter(frames), n, async_ingest=False, prefetch=False)
        assert np.array_equal(a.rec, b.rec) and a.rec.shape[0] == len(frames)
    finally:
        sgf.KLT_verbose = 1


def test_build_stream_waits_for_the_tracker_that_reads_the_slot():
    """A build on the build stream is ordered behind the tracker launches that read the slot it overwrites (per-slot read marks), not
    behind the whole main stream.  A long tracker launch (a million features) on a small frame, the slot refilled (asynchronous ingest +
    build, both far shorter than the tracker) right behind it: that tracker's result is untouched, and the next tracker sees the new
    pyramids -- the records of the same calls with a synchronisation after each (the tracker itself is pinned elsewhere)."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE
    w, h, n = 320, 240, 1000000
    base = synth.synth_base(w, h, 31)
    frames = [synth.synth_frame(w, h, 31, k, shift=(1.3, 0.8), base=base) for k in range(3)]
    tc = make_tc(levels=2, ss=4, max_residue=10.0)
    rng = np.random.default_rng(5)
    fl = np.zeros(n, FEAT_DTYPE)
    fl["x"] = rng.uniform(40, w - 40, n).astype(np.float32)
    fl["y"] = rng.uniform(40, h - 40, n).astype(np.float32)
    fl["val"] = 1
    c = Context(0)
    try:
        c.configure(tc)
        c.set_option(15, 1)
        stage = c.staging((h, w), count=1)
        stage[0][...] = frames[2]

        def run(step_sync):
            def after():
                if step_sync:
                    c.sync()
                    c.upload_wait()
            c.upload(0, frames[0])
            c.upload(1, frames[1])
            c.build_pyramids_batch([0, 1], sync=True)
            c.featbuf_upload(0, fl)
            c.sync()
            c.track_async(0, 1, 0, 1, n)                    # reads the pyramids of slots 0 and 1 ...
            after()
            c.upload_async(0, stage[0])                     # ... while slot 0 gets another frame
            after()
            c.build_pyramids(0, sync=False)                 # (build stream: must wait for the tracker above, and only for it)
            after()
            c.track_async(0, 1, 0, 2, n)
            out = c.featbuf_download(1, n), c.featbuf_download(2, n)
            c.upload_wait()
            return out

        want01, want21 = run(True)
        assert np.count_nonzero(want01["val"] >= 0) > n // 2
        assert not np.array_equal(want01["x"], want21["x"])
        for rep in range(4):
            got01, got21 = run(False)
            assert np.array_equal(got01, want01), "the tracker whose slot was refilled behind it read the new pyramids (rep %d)" % rep
            assert np.array_equal(got21, want21), "the tracker on the refilled slot did not see the new pyramids (rep %d)" % rep
    finally:
        c.close()


def test_two_rank_launch_reaches_rccl_on_one_gpu():
    """`bench.py --gpus 2` on a one-GPU box, both ranks pointed at device 0: the launcher starts two processes, they meet through
    the rendezvous file, rank 0's RCCL unique id reaches rank 1 and both call ncclCommInitRank -- where RCCL refuses two ranks
    on one device ("invalid usage").  That refusal is the proof that everything before the collective works across processes on
    the hardware; should an RCCL build accept the shared device, the two-rank line itself is checked instead."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE")}
    env["KLT_RANKS_SHARE_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "5",
                        "--resident-pairs", "4", "--inflight", "2", "--batch", "2", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode == 0:
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["parity_checked"] is True
    else:
        assert "ncclCommInitRank" in r.stderr, r.stderr[-2000:]
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("case", [
    # (width, height, window, levels, ss, min_eigenvalue, mindist, skipped pixels) -- more than 262 144 candidates each: below that a
    # replacement does not use prepared scores
    (1022, 647, 7, 2, 2, 1, 6, 0),        # a last strip that ends inside the frame's right border zone, rows that are no multiple of a tile
    (941, 701, 3, 2, 2, 1, 5, 0),         # the smallest window: 29 candidate columns per strip
    (1000, 780, 11, 2, 2, 2.5, 8, 0),     # a fractional threshold: the f32 threshold of the fused kernel against the reference's f64 compare
    (1003, 800, 15, 2, 2, 1, 10, 0),      # the reference's affine-size window: 17 candidate columns per strip, 15 rows between top and bottom
    (1100, 820, 23, 2, 2, 1, 10, 0),      # 9 candidate columns per strip
    (1100, 820, 25, 2, 2, 1, 10, 0),      # too wide for a strip: the separate column pass + eigenvalue kernels
    (1920, 1080, 7, 2, 4, 1, 10, 1),      # skipped pixels: not every pixel is a candidate -- the separate kernels
    (1920, 1080, 7, 3, 4, 1e3, 10, 0),    # a high threshold: most windows are no candidates
])
def test_prepared_scores_of_the_fused_column_and_eigenvalue_kernel(ko, case):
    """klt_select_prepare_async runs the tables' column pass and the eigenvalue keys as ONE kernel where a window fits a strip
    (sat_pipeline.hip, cols_eigen_pipe) and as the two separate kernels elsewhere; the replacement that consumes the scores gives the
    reference walk's list either way -- window sizes 3 .. 25, frames whose sides are no multiples of the strip / tile sizes, thresholds
    that are no f32 values."""
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context, REPLACING_SOME
    w, h, window, levels, ss, min_eig, mindist, skip = case
    n = max(40, w * h // 1500)       # (64 n < half the candidates: the cut that makes a replacement use prepared scores)
    f0 = synth.synth_frame(w, h, 21, 0)
    tc = make_tc(levels=levels, ss=ss, window=window, mindist=mindist, nSkippedPixels=skip)
    tc.min_eigenvalue = min_eig
    p = params_from_tc(tc)
    c = Context(0)
    try:
        c.configure(tc)
        c.set_option(15, 1)
        c.upload(0, f0)
        c.build_pyramids(0, sync=False)
        first, placed = c.select(0, n, use_pyramid=True)
        want0 = ko.select_good_features(p, f0.astype(np.float32), n)
        assert_feats(first, *oracle_feats(want0), what="selection %r" % (case,))
        if placed < 8:
            pytest.skip("too few features at this threshold to lose some")
        rng = np.random.default_rng(5)
        gone = rng.choice(np.flatnonzero(first["val"] >= 0), max(4, placed // 10), replace=False)
        fl = first.copy()
        fl["val"][gone] = -1
        fl["x"][gone] = -1.0
        fl["y"][gone] = -1.0
        want = ko.select_good_features(p, f0.astype(np.float32), n, mode=REPLACING_SOME, fl=fl.copy())
        c.select_prepare(0)
        got, _ = c.select(0, n, mode=REPLACING_SOME, fl=fl.copy(), use_pyramid=True)
        assert_feats(got, *oracle_feats(want), what="prepared replacement %r" % (case,))
        from pyfeaturetrack_amd.backend import KltBackendError
        with pytest.raises(KltBackendError, match="prepared scores"):
            c.select_intermediate(3)                    # the replacement did use the prepared scores (it wrote no eigenvalue map)
    finally:
        c.close()
