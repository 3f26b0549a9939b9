#!/usr/bin/env python3
"""Randomised differential check of the HIP path against the oracle (test infrastructure, like tests/): frame sizes of any parity,
pyramid depths, subsamplings, window sizes, minimum distances, skipped pixels, borders, residue limits and list lengths drawn at random;
per trial: pyramids of two frames, selection, tracking, replacement of the lost features on the second frame -- every record compared
bit for bit.

    python3 tests/fuzz/fuzz_parity.py [--trials 40] [--seed 1] [--max-pixels 400000] [--max-n 700] [--max-side 900]
--sequence: KLTTrackSequence against the per-frame host API loop on short random sequences (both are the HIP path; the per-frame
API is the one pinned to the reference).
Prints one line per trial and exits non-zero at the first difference (with the drawn parameters, so that it can be replayed by seed).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_tc, params_from_tc                      # noqa: E402
from pyfeaturetrack_amd import synth                             # noqa: E402
from pyfeaturetrack_amd.backend import Context, REPLACING_SOME   # noqa: E402
from oracle import klt_oracle as ko                              # noqa: E402


def same(a, b):
    return (np.array_equal(a["val"].astype(np.int64), b["val"].astype(np.int64)) and
            np.array_equal(a["x"].astype(np.float64), b["x"].astype(np.float64)) and
            np.array_equal(a["y"].astype(np.float64), b["y"].astype(np.float64)))


def draw(rng, max_pixels, max_n=700, max_side=900):
    while True:
        levels = int(rng.integers(1, 5))
        ss = int(rng.choice([2, 4, 8]))
        window = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        w = int(rng.integers(48, max_side))
        h = int(rng.integers(48, max(49, max_side * 7 // 9)))
        if w * h > max_pixels:
            continue
        # the coarsest level must hold a window and its border (the reference makes the same demand through its border arithmetic)
        coarse = ss ** (levels - 1)
        if w // coarse < window + 12 or h // coarse < window + 12:
            continue
        return dict(levels=levels, ss=ss, window=window, w=w, h=h,
                    mindist=int(rng.integers(0, 25)), skip=int(rng.integers(0, 4)),
                    smooth=bool(rng.integers(0, 2)), mr=(None if rng.random() < 0.3 else float(rng.uniform(2.0, 30.0))),
                    n=int(rng.integers(1, max_n)), seed=int(rng.integers(0, 1 << 30)),
                    shift=(float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))),
                    min_eig=int(rng.choice([1, 1, 10, 200])),
                    border=(None if rng.random() < 0.6 else int(rng.integers(window // 2 + 1, 40))),   # (smaller borders are refused: the reference reads outside the image)
                    max_iter=int(rng.choice([10, 10, 3, 25])))


def run_trial(ctx, t):
    tc = make_tc(levels=t["levels"], ss=t["ss"], window=t["window"], max_residue=t["mr"], mindist=t["mindist"],
                 nSkippedPixels=t["skip"], smoothBeforeSelecting=t["smooth"], min_eigenvalue=t["min_eig"],
                 max_iterations=t["max_iter"])
    if t["border"] is not None:
        tc.borderx = tc.bordery = t["border"]
    p = params_from_tc(tc)
    ctx.configure(tc)
    base = synth.synth_base(t["w"], t["h"], t["seed"])
    f0 = synth.shift_frame(base, 0, 0)
    f1 = synth.shift_frame(base, *t["shift"])
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids(0)
    ctx.build_pyramids(1)
    fl, _ = ctx.select(0, t["n"], use_pyramid=False)
    ofl = ko.select_good_features(p, f0.astype(np.float32), t["n"])
    if not same(fl, ofl):
        return "selection"
    out, _ = ctx.track(0, 1, fl)
    ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
    if not same(out, ofl):
        return "tracking"
    rep, _ = ctx.select(1, t["n"], mode=REPLACING_SOME, fl=out, use_pyramid=False)
    orep = ko.select_good_features(p, f1.astype(np.float32), t["n"], mode=2, fl=ofl)
    if not same(rep, orep):
        return "replacement"
    t["_stat"] = "selected %d, tracked %d, replaced %d" % (int((fl["val"] > 0).sum()), int((out["val"] == 0).sum()), int((rep["val"] > 0).sum()))
    return None


def run_prepared_trial(ctx, t):
    """The replacement pass as a sequence runs it -- scores prepared ahead (klt_select_prepare_async: row pass + cols_eigen_pipe), then
    klt_select_begin_async / klt_select_finish with REPLACING_SOME on the slot's pyramid -- against the oracle, on frames large enough for
    the prefilter cut (more than 262144 candidates) so that the passes behind the cut (four tiles per workgroup) run as well.  A random
    share of the tracked features is marked lost first, from a handful (a tight cut, sometimes its repeat) to most of the list."""
    t = dict(t, smooth=True, skip=0)
    tc = make_tc(levels=t["levels"], ss=t["ss"], window=t["window"], max_residue=t["mr"], mindist=t["mindist"],
                 nSkippedPixels=0, smoothBeforeSelecting=True, min_eigenvalue=t["min_eig"], max_iterations=t["max_iter"])
    if t["border"] is not None:
        tc.borderx = tc.bordery = t["border"]
    p = params_from_tc(tc)
    ctx.configure(tc)
    base = synth.synth_base(t["w"], t["h"], t["seed"])
    f0 = synth.shift_frame(base, 0, 0)
    f1 = synth.shift_frame(base, *t["shift"])
    ctx.upload(0, f0)
    ctx.upload(1, f1)
    ctx.build_pyramids_batch([0, 1], sync=False)
    fl, _ = ctx.select(0, t["n"], use_pyramid=True)
    ofl = ko.select_good_features(p, f0.astype(np.float32), t["n"])
    if not same(fl, ofl):
        return "selection"
    out, _ = ctx.track(0, 1, fl)
    ko.track_features(p, ko.Pyramids(p, f0.astype(np.float32)), ko.Pyramids(p, f1.astype(np.float32)), ofl)
    if not same(out, ofl):
        return "tracking"
    rng = np.random.default_rng(t["seed"])
    lose = rng.random(t["n"]) < rng.choice([0.003, 0.02, 0.2, 0.9])
    for rec in (out, ofl):
        rec["x"][lose] = -1
        rec["y"][lose] = -1
        rec["val"][lose] = -1
    ctx.select_prepare(1)
    ctx.featbuf_upload(7, out)
    ctx.select_begin(1, REPLACING_SOME, True, 7, t["n"])
    repeated = ctx.select_finish()
    rep = ctx.featbuf_download(7, t["n"])
    orep = ko.select_good_features(p, f1.astype(np.float32), t["n"], mode=2, fl=ofl)
    if not same(rep, orep):
        return "prepared replacement"
    t["_stat"] = "lost %d, replaced %d, look repeated %d" % (int(lose.sum()), int((rep["val"] > 0).sum()), int(repeated))
    return None


def run_api_trial(t, pil=False):
    """The reference-shaped Python API against the ORACLE over a random call sequence on one tracking context: KLTSelectGoodFeatures,
    KLTTrackFeatures (any two of four frames, in any order; in sequential mode frame 1 is whatever frame 2 was last time),
    KLTReplaceLostFeatures, frames edited in place between calls (a block anywhere: on or off the frame cache's lattice), a new list
    now and then -- everything the Python layer does on the way (exact frame cache with its optimistic device work, lists mapped into
    pinned memory, recycled feature objects, scores prepared ahead) must leave the reference's results.
    `pil`: the API is handed mode-"L" Pillow images (the reference's image type) -- frames 0 and 2 own their storage and are edited in place
    through Pillow (`paste`), frames 1 and 3 are mapped onto the numpy arrays the oracle reads and change with them -- all read through
    Pillow's row tables (pyfeaturetrack_amd/_pil.py)."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf, trackFeatures as tf
    sgf.KLT_verbose = tf.KLT_verbose = 0
    rng = np.random.default_rng(t["seed"])
    tc = make_tc(levels=t["levels"], ss=t["ss"], window=t["window"], max_residue=t["mr"], mindist=t["mindist"],
                 nSkippedPixels=t["skip"], smoothBeforeSelecting=t["smooth"], min_eigenvalue=t["min_eig"], max_iterations=t["max_iter"])
    if t["border"] is not None:
        tc.borderx = tc.bordery = t["border"]
    tc.sequentialMode = bool(rng.integers(0, 3) == 0)
    p = params_from_tc(tc)
    base = synth.synth_base(t["w"], t["h"], t["seed"])
    frames = [synth.synth_frame(t["w"], t["h"], t["seed"], k, shift=t["shift"], base=base) for k in range(4)]
    n = t["n"]
    colour = pil and t["seed"] % 3 == 0          # every third --pil trial: "RGB" images (the reference converts them with Pillow's luma)
    if colour:
        from PIL import Image
        rgbs = [np.dstack([f, np.roll(f, 2, axis=1), 255 - f]).astype(np.uint8) for f in frames]
        imgs = [Image.frombytes("RGB", (t["w"], t["h"]), c.tobytes()) if k % 2 == 0 else Image.fromarray(c, "RGB") for k, c in enumerate(rgbs)]
        frames = [np.array(im.convert("F")) for im in imgs]      # what the oracle is given: the reference's own conversion of these images
    elif pil:
        from PIL import Image
        imgs = [Image.frombytes("L", (t["w"], t["h"]), f.tobytes()) if k % 2 == 0 else Image.fromarray(f) for k, f in enumerate(frames)]
    else:
        imgs = frames

    def same_list(fl, ofl, what):
        have = np.array([(f.x, f.y, f.val) for f in fl], np.float64).reshape(-1, 3)
        ok = (np.array_equal(have[:, 0], ofl["x"].astype(np.float64)) and np.array_equal(have[:, 1], ofl["y"].astype(np.float64))
              and np.array_equal(have[:, 2], ofl["val"].astype(np.float64)))
        return None if ok else what

    cur = int(rng.integers(0, 4))
    fl = sgf.KLTSelectGoodFeatures(tc, imgs[cur], n)
    ofl = ko.select_good_features(p, frames[cur].astype(np.float32), n)
    bad = same_list(fl, ofl, "op 0: select")
    last2 = None                                 # pixels of the last call's frame 2 (what sequential mode tracks from)
    ops = []
    for step in range(1, 9):
        if bad:
            break
        op = rng.choice(["track", "track", "track", "replace", "edit", "select"])
        ops.append(op)
        if op == "edit":
            k = int(rng.integers(0, 4))
            y, x = int(rng.integers(0, t["h"] - 8)), int(rng.integers(0, t["w"] - 8))
            hh, ww = int(rng.integers(1, 8)), int(rng.integers(1, 8))
            if colour:                              # one channel of a block: the array-mapped image changes with its array, the other through Pillow
                rgbs[k][y:y + hh, x:x + ww, int(rng.integers(0, 3))] ^= int(rng.integers(1, 255))
                if k % 2 == 0:
                    imgs[k].paste(Image.fromarray(rgbs[k][y:y + hh, x:x + ww].copy(), "RGB"), (x, y))
                frames[k] = np.array(imgs[k].convert("F"))
                continue
            frames[k][y:y + hh, x:x + ww] ^= int(rng.integers(1, 255))
            if pil and k % 2 == 0:                  # an image with storage of its own: the same edit through Pillow, on the same object
                imgs[k].paste(Image.fromarray(frames[k][y:y + hh, x:x + ww].copy()), (x, y))
        elif op == "select":
            cur = int(rng.integers(0, 4))
            fl = sgf.KLTSelectGoodFeatures(tc, imgs[cur], n)
            ofl = ko.select_good_features(p, frames[cur].astype(np.float32), n)
            bad = same_list(fl, ofl, "op %d: select" % step)
        elif op == "replace":
            sgf.KLTReplaceLostFeatures(tc, imgs[cur], fl)
            if int((ofl["val"] < 0).sum()) > 0:
                if tc.sequentialMode and last2 is not None:
                    # selectGoodFeatures.py:176-181: level 0 of the pyramids kept from the last track -- the SMOOTHED frame 2 and its
                    # gradients, whatever smoothBeforeSelecting says
                    import copy
                    q = copy.copy(p)
                    q.smoothBeforeSelecting = 1
                    ofl = ko.select_good_features(q, last2.astype(np.float32), n, mode=2, fl=ofl)
                else:
                    ofl = ko.select_good_features(p, frames[cur].astype(np.float32), n, mode=2, fl=ofl)
            bad = same_list(fl, ofl, "op %d: replace" % step)
        else:
            nxt = int(rng.integers(0, 4))
            tf.KLTTrackFeatures(tc, imgs[cur], imgs[nxt], fl)
            first = last2 if (tc.sequentialMode and last2 is not None) else frames[cur]          # trackFeatures.py:152-161
            ko.track_features(p, ko.Pyramids(p, first.astype(np.float32)), ko.Pyramids(p, frames[nxt].astype(np.float32)), ofl)
            last2 = frames[nxt].copy()
            cur = nxt
            bad = same_list(fl, ofl, "op %d: track" % step)
    t["_stat"] = "%ssequential %d, ops %s, alive at the end %d" % ("RGB images, " if colour else "", tc.sequentialMode, "".join(o[0] for o in ops), int((ofl["val"] >= 0).sum()))
    return bad


def run_sequence_trial(t):
    """KLTTrackSequence (device-resident table, build stream, prepared scores, frame stager) against the per-frame host API loop it
    replaces (track, replace, store) on 5-7 frames; one region of one frame is wiped so that features are lost and replaced."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf, storeFeatures as sf, trackFeatures as tf
    from pyfeaturetrack_amd.klt import KLT_TrackingContext
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    sgf.KLT_verbose = tf.KLT_verbose = 0
    nf = 5 + t["seed"] % 3
    base = synth.synth_base(t["w"], t["h"], t["seed"])
    frames = [synth.synth_frame(t["w"], t["h"], t["seed"], k, shift=t["shift"], base=base) for k in range(nf)]
    frames[2] = frames[2].copy()
    frames[2][t["h"] // 4:t["h"] // 2, t["w"] // 4:t["w"] // 2] = 100

    def make():
        tc = KLT_TrackingContext()
        tc.window_width = tc.window_height = t["window"]
        tc.nPyramidLevels, tc.subsampling = t["levels"], t["ss"]
        tc.KLTUpdateTCBorder()
        tc.mindist, tc.nSkippedPixels, tc.smoothBeforeSelecting = t["mindist"], t["skip"], t["smooth"]
        tc.max_residue, tc.min_eigenvalue, tc.max_iterations = t["mr"], t["min_eig"], t["max_iter"]
        tc.sequentialMode = True
        return tc

    replace, ingest, prefetch = bool(t["seed"] & 8), bool(t["seed"] & 16), bool(t["seed"] & 32)
    tc = make()
    want = sf.KLTCreateFeatureTable(nf, t["n"])
    fl = sgf.KLTSelectGoodFeatures(tc, frames[0], t["n"])
    sf.KLTStoreFeatureList(fl, want, 0)
    for k in range(1, nf):
        tf.KLTTrackFeatures(tc, frames[k - 1], frames[k], fl)
        if replace:
            sgf.KLTReplaceLostFeatures(tc, frames[k], fl)
        sf.KLTStoreFeatureList(fl, want, k)
    got = KLTTrackSequence(make(), (f for f in frames), t["n"], replace_lost=replace, async_ingest=ingest, prefetch=prefetch)
    t["_stat"] = "frames %d, replace %d, ingest %d, prefetch %d, lost entries %d, replaced %d" % (
        nf, replace, ingest, prefetch, int((want.val[1:] < 0).sum()), int((want.val[1:] > 0).sum()))
    if not (np.array_equal(got.val, want.val) and np.array_equal(got.x, want.x) and np.array_equal(got.y, want.y)):
        bad = np.argwhere((got.val != want.val) | (got.x != want.x) | (got.y != want.y))
        t["_diff"] = "%d entries differ, first (frame, feature) %s: per-frame API (%g, %g, %d), KLTTrackSequence (%g, %g, %d)" % (
            len(bad), tuple(bad[0]), want.x[tuple(bad[0])], want.y[tuple(bad[0])], want.val[tuple(bad[0])],
            got.x[tuple(bad[0])], got.y[tuple(bad[0])], got.val[tuple(bad[0])])
        return "sequence table"
    return None


def run_affine_trial(ctx, t):
    """The affine consistency check (parity unpinned: the reference does not define it) -- HIP against the oracle on a texture warped
    by a random affine map per frame: every status, position, template offset and matrix entry after each of three calls."""
    from pyfeaturetrack_amd.params import affine_params_from_tc
    rng = np.random.default_rng(t["seed"])
    tc = make_tc(levels=t["levels"], ss=t["ss"], window=t["window"], max_residue=t["mr"], mindist=t["mindist"],
                 nSkippedPixels=t["skip"], min_eigenvalue=t["min_eig"], max_iterations=t["max_iter"])
    tc.affineConsistencyCheck = int(rng.integers(0, 3))
    tc.affine_window_width = tc.affine_window_height = int(rng.choice([9, 11, 15, 15, 21]))
    tc.affine_max_residue = float(rng.uniform(4.0, 20.0))
    tc.affine_max_iterations = int(rng.choice([3, 10, 10, 20]))
    tc.affine_max_displacement_differ = float(rng.choice([0.3, 1.5, 1.5, 5.0]))
    p, ap = params_from_tc(tc), affine_params_from_tc(tc)
    ang, sc = float(rng.uniform(-1.0, 1.0)), float(rng.uniform(0.995, 1.005))
    c, s_ = np.cos(np.radians(ang)) * sc, np.sin(np.radians(ang)) * sc
    step = np.array([[c, -s_], [s_, c]]) + rng.uniform(-0.003, 0.003, (2, 2)) * (tc.affineConsistencyCheck == 2)
    base = synth.synth_base(t["w"], t["h"], t["seed"], sigma=2.5)
    frames, A = [], np.eye(2)
    for k in range(4):
        frames.append(synth.warp_frame(base, A, (k * t["shift"][0] * 0.4, k * t["shift"][1] * 0.4)))
        A = step @ A
    ctx.configure(tc)
    for k, f in enumerate(frames):
        ctx.upload(k, f)
    ctx.build_pyramids_batch(list(range(4)), sync=True)
    n = t["n"]
    fl, _ = ctx.select(0, n, use_pyramid=True)
    f32 = [f.astype(np.float32) for f in frames]
    P = [ko.Pyramids(p, f) for f in f32]
    # the oracle's list starts from the HIP selection (selection on the smoothed level-0 image is the sequence API's; compared elsewhere)
    ofl = fl.copy()
    ctx.affine_alloc(0, n)
    st = ko.AffineState(ap, n)
    live = 0
    try:
        for k in range(1, 4):
            fl, _ = ctx.track_affine(k - 1, k, fl, 0)
            ko.track_features_affine(p, P[k - 1], P[k], ofl, st)
            rec = ctx.affine_download(0, n)
            if not same(fl, ofl):
                return "affine call %d: records" % k
            for name in ("valid", "aff_x", "aff_y", "Axx", "Ayx", "Axy", "Ayy"):
                if not np.array_equal(rec[name], st.rec[name]):
                    return "affine call %d: %s" % (k, name)
            live = int((fl["val"] == 0).sum())
    finally:
        ctx.affine_free(0)              # (the affine window of the next draw differs: no state may exist when it is set)
    t["_stat"] = "mode %d, window %d, %d of %d alive after three calls" % (tc.affineConsistencyCheck, tc.affine_window_width, live, n)
    return None


def run_batch_trial(ctx, t):
    """2-6 pairs of one random size through the batched calls (one pyramid build for all frames, one tracker launch for all pairs, feature
    lists as views of one table) against the oracle pair by pair."""
    rng = np.random.default_rng(t["seed"])
    B, n = int(rng.integers(2, 7)), t["n"]
    tc = make_tc(levels=t["levels"], ss=t["ss"], window=t["window"], max_residue=t["mr"], mindist=t["mindist"],
                 nSkippedPixels=t["skip"], min_eigenvalue=t["min_eig"], max_iterations=t["max_iter"])
    p = params_from_tc(tc)
    ctx.configure(tc)
    pairs = []
    for b in range(B):
        base = synth.synth_base(t["w"], t["h"], t["seed"] + b)
        sh = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        pairs.append((synth.shift_frame(base, 0, 0), synth.shift_frame(base, *sh)))
        ctx.upload(2 * b, pairs[b][0])
        ctx.upload(2 * b + 1, pairs[b][1])
    ctx.build_pyramids_batch(list(range(2 * B)))
    T_IN, T_OUT, V_IN, V_OUT = 400, 401, 410, 420
    ctx.featbuf_alloc(T_IN, B * n)
    ctx.featbuf_alloc(T_OUT, B * n)
    for b in range(B):
        ctx.featbuf_view(V_IN + b, T_IN, b * n, n)
        ctx.featbuf_view(V_OUT + b, T_OUT, b * n, n)
        ctx.select_async(2 * b, 1, True, V_IN + b, n)            # SELECTING_ALL on the smoothed level-0 image
    ctx.track_batch_async([(2 * b, 2 * b + 1, V_IN + b, V_OUT + b) for b in range(B)], n)
    sel = ctx.featbuf_download(T_IN, B * n).reshape(B, n)
    out = ctx.featbuf_download(T_OUT, B * n).reshape(B, n)
    tracked = 0
    for b in range(B):
        P0, P1 = ko.Pyramids(p, pairs[b][0].astype(np.float32)), ko.Pyramids(p, pairs[b][1].astype(np.float32))
        ofl = sel[b].copy()                                         # (the selection itself is compared by the plain trials)
        ko.track_features(p, P0, P1, ofl)
        if not same(out[b], ofl):
            return "pair %d of %d" % (b, B)
        tracked += int((ofl["val"] == 0).sum())
    t["_stat"] = "%d pairs, tracked %d of %d" % (B, tracked, B * n)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-pixels", type=int, default=400000)
    ap.add_argument("--max-n", type=int, default=700)
    ap.add_argument("--max-side", type=int, default=900)
    ap.add_argument("--batch", action="store_true", help="batched build + one tracker launch for several pairs, against the oracle")
    ap.add_argument("--affine", action="store_true", help="the affine consistency check, HIP against the oracle")
    ap.add_argument("--sequence", action="store_true", help="KLTTrackSequence against the per-frame host API instead of HIP against the oracle")
    ap.add_argument("--prepared", action="store_true", help="replacement on prepared scores (klt_select_prepare_async + begin / finish), HIP against the oracle")
    ap.add_argument("--api", action="store_true", help="the reference-shaped Python API over random call sequences, against the oracle")
    ap.add_argument("--pil", action="store_true", help="with --api: the calls are handed Pillow images (owned storage and array-mapped ones), edited in place")
    ap.add_argument("--min-pixels", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = None if (a.sequence or a.api) else Context(0)
    t0 = time.time()
    for k in range(a.trials):
        t = draw(rng, a.max_pixels, a.max_n, a.max_side)
        while t["w"] * t["h"] < a.min_pixels:
            t = draw(rng, a.max_pixels, a.max_n, a.max_side)
        try:
            bad = run_api_trial(t, a.pil) if a.api else run_prepared_trial(ctx, t) if a.prepared else run_sequence_trial(t) if a.sequence else run_affine_trial(ctx, t) if a.affine else run_batch_trial(ctx, t) if a.batch else run_trial(ctx, t)
        except SystemExit as e:            # KLTError of the host layer
            bad = "error: %s" % (e,)
        print("trial %3d %s  %s" % (k, "ok  " if not bad else "FAIL (%s)" % bad, t), flush=True)
        if bad:
            sys.exit(1)
    print("%d trials identical in %.0f s" % (a.trials, time.time() - t0))
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
