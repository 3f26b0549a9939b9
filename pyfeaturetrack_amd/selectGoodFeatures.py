"""KLTSelectGoodFeatures (reference: selectGoodFeatures.py) on the MI355X backend.

The reference smooths the frame, differentiates it, scores every interior pixel with the minimum
eigenvalue of the windowed gradient matrix (goodFeaturesUtils.pyx:35-73), sorts ~W*H Python tuples
and walks them greedily (selectGoodFeatures.py:230-251).  Here the frame goes to the device once and
klt_select runs all of that as HIP kernels; only the n selected 16-byte records come back.
"""
from __future__ import print_function

import numpy as np

from .backend import FEAT_DTYPE, REPLACING_SOME, SELECTING_ALL, context_of, default_context  # noqa: F401
from .klt import KLT_Feature, KLTCountRemainingFeatures, kltState, new_feature_list, shared_store
from .error import KLTWarning


class selectionMode:
    SELECTING_ALL = SELECTING_ALL
    REPLACING_SOME = REPLACING_SOME


KLT_verbose = 1


def image_to_array(img):
    """PIL image (or ndarray) -> uint8 or float32 2-D array holding exactly `np.array(img.convert("F"))`."""
    if isinstance(img, np.ndarray):
        return img if img.dtype == np.uint8 else np.ascontiguousarray(img, np.float32)
    if getattr(img, "mode", None) == "L":
        return np.asarray(img, np.uint8)          # convert("F") of an 8-bit image is the exact u8 value
    return np.array(img.convert("F"), np.float32)


def _image_size(img):
    if isinstance(img, np.ndarray):
        return img.shape[1], img.shape[0]
    return img.size


_UNKNOWN = object()


def features_to_array(featurelist, out=None, store=_UNKNOWN):
    """KLT_Feature list -> record array (`out`, or a new one).  A list whose features share one column store (every list this
    package hands out) is converted column by column; anything else feature by feature.  `store`: what shared_store(featurelist)
    gave, for a caller that has asked already."""
    fl = np.zeros(len(featurelist), FEAT_DTYPE) if out is None else out
    if out is not None:
        fl["aux"] = 0
    if store is _UNKNOWN:
        store = shared_store(featurelist)
    if store is not None:
        fl["x"], fl["y"], fl["val"] = store.x, store.y, store.val
    elif len(featurelist):
        fl["x"] = [f.x for f in featurelist]
        fl["y"] = [f.y for f in featurelist]
        fl["val"] = [f.val for f in featurelist]
    return fl


def _slots_of(tc):
    """Two device slots per tracking context: [frame 1, frame 2]; a third for selection frames."""
    s = tc.__dict__.get("_klt_slots")
    if s is None:
        import weakref
        ctx = context_of(tc)
        base = ctx.take_slots(3)
        s = tc._klt_slots = (base, base + 1, base + 2)
        weakref.finalize(tc, ctx.release_slots, base, 3)      # the device memory goes when the tracking context does
    return s


def _pyramid_fits(tc, ncols, nrows):
    """every level of the tracking context's pyramid has at least one pixel (pyramid.py:26-31: int(n / ss) per level)"""
    for _ in range(int(tc.nPyramidLevels) - 1):
        ncols, nrows = ncols // int(tc.subsampling), nrows // int(tc.subsampling)
    return ncols >= 1 and nrows >= 1


def _fix_window(tc):
    if tc.window_width % 2 != 1:
        tc.window_width += 1
        KLTWarning("Tracking context's window width must be odd.  Changing to {0}.\n".format(tc.window_width))
    if tc.window_height % 2 != 1:
        tc.window_height += 1
        KLTWarning("Tracking context's window height must be odd.  Changing to {0}.\n".format(tc.window_height))
    if tc.window_width < 3:
        tc.window_width = 3
        KLTWarning("Tracking context's window width must be at least three.  \nChanging to 3.\n")
    if tc.window_height < 3:
        tc.window_height = 3
        KLTWarning("Tracking context's window height must be at least three.  \nChanging to 3.\n")


def _fillFeaturemap(x, y, featuremap, mindist, ncols, nrows):
    """selectGoodFeatures.py:18-25: marks the (2 mindist + 1)-square around (x, y), clipped to the image, in the flat row-major
    `featuremap` (a list of bools, or any flat sequence that takes item assignment) and returns it."""
    x0, x1 = max(x - mindist, 0), min(x + mindist, ncols - 1)
    if x0 <= x1:
        for iy in range(max(y - mindist, 0), min(y + mindist, nrows - 1) + 1):
            for k in range(iy * ncols + x0, iy * ncols + x1 + 1):
                featuremap[k] = True
    return featuremap


def _enforceMinimumDistance(pointlist, featurelist, ncols, nrows, mindist, min_eigenvalue, overwriteAllFeatures):
    """selectGoodFeatures.py:45-135 under the reference's own name: `pointlist` = [(val, x, y), ...] is walked IN THE ORDER GIVEN (the
    reference's caller has sorted it, :234-236); a point is placed into the next free slot of `featurelist` -- every slot in turn when
    `overwriteAllFeatures`, the lost ones (val < 0) otherwise -- unless it lies within mindist - 1 (Chebyshev) of a feature placed
    before it or, when the old features are kept, of a live one, or its value is below max(min_eigenvalue, 1).  The walk itself runs
    on the device (klt_min_distance_walk); the points it would skip without effect are dropped here first.  Returns the list.
    Values are C floats as ScanImageForGoodFeatures produces them; a point outside the image is the reference's AssertionError."""
    if min_eigenvalue < 1:
        min_eigenvalue = 1
    n = len(featurelist)
    pts = np.asarray([(p[0], p[1], p[2]) for p in pointlist], np.float64).reshape(-1, 3)
    val, px, py = pts[:, 0], pts[:, 1].astype(np.int64), pts[:, 2].astype(np.int64)
    if n == 0:
        return featurelist
    d = mindist - 1
    store = shared_store(featurelist)
    rec = features_to_array(featurelist, None, store)
    keep = val >= min_eigenvalue
    if not overwriteAllFeatures and d >= 0:
        seeded = np.zeros((nrows, ncols), bool)
        for i in np.nonzero(rec["val"] >= 0)[0]:
            x, y = int(rec["x"][i]), int(rec["y"][i])           # int(feat.x), int(feat.y): truncation (:66-67)
            seeded[max(y - d, 0):max(min(y + d, nrows - 1) + 1, 0), max(x - d, 0):max(min(x + d, ncols - 1) + 1, 0)] = True
    else:
        seeded = None
    # The reference asserts a point's bounds when the walk reaches it (:100-103) and stops looking once the list is full; a point
    # list with an out-of-image entry behind that point is accepted there and refused here -- the stricter reading.
    assert bool(np.all((px >= 0) & (px < ncols) & (py >= 0) & (py < nrows))), "point outside the image"
    if seeded is not None and len(px):
        keep &= ~seeded[py, px]
    v32 = val[keep].astype(np.float32)
    keys = (v32.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (px[keep].astype(np.uint64) << np.uint64(16)) | py[keep].astype(np.uint64)
    ctx = default_context()
    with ctx.lock:
        out, placed = ctx.min_distance_walk(keys, ncols, nrows, mindist, overwriteAllFeatures, rec)
    old = rec["val"]
    if overwriteAllFeatures:
        # slots the walk reached are rewritten by rank; behind them only the lost ones become (-1, -1, KLT_NOT_FOUND) (:80-82)
        touched = np.arange(n) < placed
        touched |= old < 0
    else:
        touched = (old < 0) & (out["val"] >= 0)
    if store is not None:
        np.copyto(store.x, out["x"], where=touched)
        np.copyto(store.y, out["y"], where=touched)
        np.copyto(store.val, out["val"], where=touched)
        np.logical_or(store.xint, touched, out=store.xint)
        np.logical_or(store.yint, touched, out=store.yint)
        store.reset_affine(touched)
        store.changed()
        return featurelist
    xs, ys, vs = out["x"].tolist(), out["y"].tolist(), out["val"].tolist()
    for i in np.nonzero(touched)[0]:
        feat = featurelist[i]
        feat.x, feat.y, feat.val = int(xs[i]), int(ys[i]), vs[i]
        feat.aff_img = feat.aff_img_gradx = feat.aff_img_grady = None
        feat.aff_x, feat.aff_y = -1.0, -1.0
        feat.aff_Axx, feat.aff_Ayx, feat.aff_Axy, feat.aff_Ayy = 1.0, 0.0, 0.0, 1.0
    return featurelist


def _KLTSelectGoodFeatures(tc, img, nFeatures, mode, featurelist=None):
    """selectGoodFeatures.py:141-261.  With REPLACING_SOME the given list is updated in place."""
    _fix_window(tc)
    if tc.mindist < 0:
        KLTWarning("(_KLTSelectGoodFeatures) Tracking context field tc.mindist is negative ({0}); setting to zero".format(tc.mindist))
        tc.mindist = 0
    ctx = context_of(tc)
    with ctx.lock:                                      # one KLT* call at a time per device context (backend.default_context)
        ctx.settle_deferred()
        return _select_locked(ctx, tc, img, nFeatures, mode, featurelist)


def _select_locked(ctx, tc, img, nFeatures, mode, featurelist):
    from ._frames import FrameKey, cache_of, settle_frames
    if cache_of(tc).handles and not ctx.configured_for(tc):
        cache_of(tc).keep_all_handles()               # new parameters void every pyramid of the context: kept handles fetch theirs first
    ctx.configure(tc)
    slots = _slots_of(tc)
    reuse = (mode == selectionMode.REPLACING_SOME and tc.sequentialMode and tc.pyramid_last is not None
             and ctx.pyramids_valid(slots[0]))
    if reuse:
        slot = slots[0]            # selectGoodFeatures.py:176-181: level 0 of the pyramids kept from the last track
    elif tc.smoothBeforeSelecting and _pyramid_fits(tc, *_image_size(img)):
        # The smoothed image and its gradients ARE level 0 of the image's pyramids (same taps: selectGoodFeatures.py:183-197 and
        # trackFeatures.py:165-172).  An image one of the two frame slots already holds (_frames.py) is neither uploaded nor
        # smoothed again; a new one goes to the slot the next KLTTrackFeatures call will look for it in (frame 1 -- or frame 2 while
        # sequential mode keeps the last frame in slot 1) and gets its whole pyramid, which that call then does not build.
        frames = cache_of(tc)
        key = FrameKey(img)
        slot = frames.find(key, slots[:2], ctx)
        if slot is None:
            slot = slots[1] if (tc.sequentialMode and tc.pyramid_last is not None) else slots[0]
            frames.send(ctx, slot, key)
            ctx.build_pyramids(slot, sync=False)
        elif not ctx.pyramids_valid(slot):
            frames.keep_handles(ctx, slot)
            ctx.build_pyramids(slot, sync=False)
        reuse = True
    else:
        # gradients of the raw frame (nothing a pyramid holds) -- or a frame too small for the tracking context's pyramid, e.g. after
        # KLTChangeTCPyramid with a large search range: the reference's selection never builds a pyramid (selectGoodFeatures.py:183-197)
        # and succeeds on such a frame, so the selection smooths and differentiates in its own slot
        slot = slots[2]
        from ._frames import pixels_of
        ctx.upload(slot, pixels_of(img))
    replacing = mode == selectionMode.REPLACING_SOME
    n = int(nFeatures) if featurelist is None else len(featurelist)
    store = fl_in = aff = was_lost = None
    if featurelist is not None:
        store = shared_store(featurelist)
        if replacing:
            fl_in = features_to_array(featurelist, ctx.host_records(n)[0], store)
            was_lost = fl_in["val"] < 0                 # (before the device updates that array in place)
        from .trackFeatures import affine_state_lookup
        aff = affine_state_lookup(ctx, featurelist)
    if aff is not None and aff[1] == n:
        ctx.set_option(4, aff[0])       # newly placed features lose their affine templates (:120-128)
    try:
        ctx.select_enqueue(slot, n, mode=mode, use_pyramid=reuse)
        try:
            if featurelist is None:
                # the reference's `[KLT_Feature() for i in range(nFeatures)]` (:143), made while the device scores, sorts and picks
                # (or taken over from a list the caller has dropped: klt._recycled)
                featurelist = new_feature_list(n)
                store = featurelist._store
        finally:
            fl = ctx.select_complete(n)
    finally:
        if aff is not None:
            ctx.set_option(4, -1)
    settle_frames(ctx)
    affine_used = aff is not None or tc.affineConsistencyCheck >= 0     # otherwise the affine fields were never assigned
    vals = fl["val"]
    if store is not None:
        # whole columns at once (selectGoodFeatures.py:109-128 touches every feature object in a Python loop)
        if replacing:
            placed = was_lost                                                 # live features are left untouched (:109-110)
            np.logical_and(placed, vals >= 0, out=placed)
            touched = placed
            np.copyto(store.x, fl["x"], where=placed)                         # integer positions (:116-119)
            np.copyto(store.y, fl["y"], where=placed)
            np.copyto(store.val, vals, where=placed)
            np.logical_or(store.xint, placed, out=store.xint)
            np.logical_or(store.yint, placed, out=store.yint)
        else:
            # every slot is written: the placed features, then (-1, -1, KLT_NOT_FOUND) where the candidates ran out
            touched = slice(None)
            store.x[:] = fl["x"]
            store.y[:] = fl["y"]
            store.val[:] = vals
            store.xint[:] = True
            store.yint[:] = True
        if affine_used:
            store.reset_affine(touched)
        store.changed()
        return featurelist
    xs, ys, vals = fl["x"].tolist(), fl["y"].tolist(), vals.tolist()
    lost = was_lost.tolist() if replacing else None
    for i, feat in enumerate(featurelist):
        if lost is not None and not lost[i]:
            continue                # live features are left untouched (:109-110)
        if vals[i] >= 0:
            feat.x = int(xs[i])
            feat.y = int(ys[i])
            feat.val = vals[i]
        elif mode == selectionMode.SELECTING_ALL:
            feat.x = -1
            feat.y = -1
            feat.val = kltState.KLT_NOT_FOUND
        else:
            continue
        if affine_used:
            feat._reset_affine()
    return featurelist


def KLTSelectGoodFeatures(tc, img, nFeatures):
    """selectGoodFeatures.py:279-294"""
    ncols, nrows = _image_size(img)
    if KLT_verbose >= 1:
        print("(KLT) Selecting the {0} best features from a {1} by {2} image...  ".format(nFeatures, ncols, nrows))
    fl = _KLTSelectGoodFeatures(tc, img, nFeatures, selectionMode.SELECTING_ALL)
    if KLT_verbose >= 1:
        print("\n\t{0} features found.\n".format(KLTCountRemainingFeatures(fl)))
    return fl


def KLTReplaceLostFeatures(tc, img, featurelist):
    """Upstream KLT's KLTReplaceLostFeatures (absent from the reference, which only carries the
    REPLACING_SOME plumbing: selectGoodFeatures.py:64-69, :109-110, :176-181).  Lost features
    (val < 0) are replaced by the best new candidates that keep `mindist` to every live feature."""
    ncols, nrows = _image_size(img)
    nLost = len(featurelist) - KLTCountRemainingFeatures(featurelist)
    if KLT_verbose >= 1:
        print("(KLT) Attempting to replace {0} features in a {1} by {2} image...  ".format(nLost, ncols, nrows))
    tc.__dict__["_klt_replaced_after_track"] = True        # (KLTTrackFeatures then prepares the next replacement's scores ahead)
    if nLost > 0:
        _KLTSelectGoodFeatures(tc, img, len(featurelist), selectionMode.REPLACING_SOME, featurelist)
    if KLT_verbose >= 1:
        print("\n\t{0} features replaced.".format(nLost - len(featurelist) + KLTCountRemainingFeatures(featurelist)))
