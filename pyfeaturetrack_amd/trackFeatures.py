"""KLTTrackFeatures (reference: trackFeatures.py) on the MI355X backend.

One call = upload the frame(s) as u8, build the image / gradx / grady pyramids of each on the
device (ComputeImagePyramids, trackFeatures.py:146-196) and run one tracker launch over all live
features (one wavefront per feature, all pyramid levels inside the kernel).  In sequentialMode the
frame-2 pyramids stay resident and become frame 1 of the next call (:152-161, :401-404).
"""
from __future__ import print_function

from . import selectGoodFeatures as _sgf
import numpy as np

from ._abi import KltBackendError
from .backend import context_of, default_context  # noqa: F401
from .klt import KLTCountRemainingFeatures, kltState, shared_store  # noqa: F401
from ._frames import FrameKey, KLTForgetFrames, cache_of, settle_frames  # noqa: F401
from .selectGoodFeatures import _fix_window, _image_size, _slots_of, features_to_array, image_to_array
# the reference binds the name at import (trackFeatures.py:7 `from selectGoodFeatures import KLT_verbose`): this module has its own
# switch, and setting selectGoodFeatures.KLT_verbose later does not reach it
from .selectGoodFeatures import KLT_verbose  # noqa: E402


class _Levels:
    """`pyramid.img` of a device-resident pyramid: a sequence of the levels' float32 planes, each downloaded when first looked at.
    It carries everything a download needs (context, the tracking context's slots, the generation of the build it stands for) and
    no reference back to the pyramid object: a pyramid handle that is dropped -- `tc.pyramid_last*` are replaced on every
    sequential-mode call -- is freed at once and never fetches its planes (with a cycle here the dropped handles of the last call
    were still alive when their slot was overwritten, and every call downloaded nine planes nobody would look at: 4.7 instead of
    0.5 ms per 1080p frame).  These objects are what the frame cache watches (`FrameCache.watch`)."""

    def __init__(self, ctx, slots, gen, plane, nlevels):
        self._ctx, self._slots, self._gen, self._plane = ctx, tuple(slots), gen, plane
        self._planes = [None] * nlevels

    def __len__(self):
        return len(self._planes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self._planes)))]
        if self._planes[i] is None:
            self._fetch(range(len(self._planes))[i])
        return self._planes[i]

    def __setitem__(self, i, plane):
        self._planes[i] = plane

    def __iter__(self):
        return (self[i] for i in range(len(self._planes)))

    def _slot(self):
        for s in self._slots:
            if self._ctx.slot_generation(s) == self._gen:
                return s
        from ._abi import KltBackendError
        raise KltBackendError("the pyramid planes this handle stands for have been replaced on the device by a call outside the "
                              "KLT* functions of this package (klt_build_pyramids on the tracking context's slot)")

    def _fetch(self, level):
        self._planes[level] = self._ctx.download_level(self._slot(), self._plane, level)

    def _materialise(self):
        for level in range(len(self._planes)):
            if self._planes[level] is None:
                self._fetch(level)


class _ResidentPyramids:
    """A KLTPyramid (pyramid.py:15-35: subsampling, nLevels, ncols[], nrows[], img[]) whose planes live in a device slot -- what
    ComputeImagePyramids returns and what tc.pyramid_last* point at in sequential mode.  `img[i]` downloads level i on first access;
    before the API layer overwrites the slot (a new frame, a rebuild) it fetches the levels of every handle that is still alive, so
    a handle somebody kept stays valid as the reference's pyramid objects do."""

    def __init__(self, ctx, slots, gen, which, ncols, nrows, subsampling, nlevels):
        self.which = which
        self.subsampling, self.nLevels = subsampling, nlevels
        self.ncols, self.nrows = [], []
        for _ in range(nlevels):                  # true division, as in the reference (pyramid.py:26-31: levels >= 1 hold floats)
            self.ncols.append(ncols)
            self.nrows.append(nrows)
            ncols, nrows = ncols / subsampling, nrows / subsampling
        self.img = _Levels(ctx, slots, gen, ("img", "gradx", "grady").index(which), nlevels)

    _gen = property(lambda self: self.img._gen)

    def _materialise(self):
        self.img._materialise()

    def __repr__(self):
        return "<device pyramid %s, %dx%d, %d levels>" % (self.which, self.ncols[0], self.nrows[0], self.nLevels)


def _pyramid_handles(tc, ctx, slot, ncols, nrows):
    """(image, gradx, grady) handles of the pyramids in `slot`, registered with the tracking context's frame cache"""
    gen = ctx.slot_generation(slot)
    hs = tuple(_ResidentPyramids(ctx, _slots_of(tc), gen, which, ncols, nrows, int(tc.subsampling), tc.nPyramidLevels)
               for which in ("img", "gradx", "grady"))
    cache_of(tc).watch(h.img for h in hs)           # (the plane holders: they live as long as somebody holds the pyramid or its img)
    return hs


_DEVICE_TEMPLATE = "<template on device>"      # what feat.aff_img* hold while the device keeps the templates


def _anchor(featurelist):
    """the object whose life stands for the list's features': the column store of the first feature (KLT_Feature objects are
    (store, row) pairs and, like any tuple, cannot be weakly referenced; the store lives exactly as long as one of its features
    does), or the first element itself for a list of foreign objects"""
    first = featurelist[0]
    return getattr(first, "_s", first)


class _ListRef:
    """Weak handle on a feature list (plain lists cannot be weakly referenced): the first feature's column store stands in for it.
    The reference keeps the affine state in the KLT_Feature objects themselves, so the state dies with them."""
    def __init__(self, featurelist):
        import weakref
        self.ident = id(featurelist)
        self.first = weakref.ref(_anchor(featurelist)) if len(featurelist) else None

    def matches(self, featurelist):
        return (self.first is not None and len(featurelist) > 0 and self.first() is _anchor(featurelist)
                and self.ident == id(featurelist))


def affine_state_lookup(ctx, featurelist):
    """(state id, length) of the device affine state that travels with this feature list, or None."""
    entry = ctx.__dict__.setdefault("_affine_states", {}).get(id(featurelist))
    if entry is None or not entry[0].matches(featurelist):
        return None
    return entry[1], entry[2]


def _affine_state_of(tc, ctx, featurelist):
    """Device affine-state id of a feature list (allocated on first use).  The table is keyed by the list's id but every entry
    carries a weak reference to the list's first feature: an id that CPython re-uses for another list never inherits stale
    templates, and the device state is released when the features die."""
    import weakref
    table = ctx.__dict__.setdefault("_affine_states", {})
    key = id(featurelist)
    entry = table.get(key)
    if entry is not None and (not entry[0].matches(featurelist) or entry[2] != len(featurelist)):
        ctx.release_affine_state(entry[1])
        entry = None
    if entry is None:
        sid = ctx.take_affine_state()
        ctx.affine_alloc(sid, len(featurelist))
        entry = table[key] = (_ListRef(featurelist), sid, len(featurelist))
        if len(featurelist):
            def _drop(table=table, key=key, sid=sid, ctx=ctx):
                if key in table and table[key][1] == sid:
                    del table[key]
                    ctx.release_affine_state(sid)
            anchor = _anchor(featurelist)
            if hasattr(anchor, "when_features_die"):
                anchor.when_features_die(_drop)          # (also when the objects are handed to a new list: klt._recycled)
            else:
                weakref.finalize(anchor, _drop)
    return entry[1]


def _trackFeature(x1, y1, x2, y2, img1, gradx1, grady1, img2, gradx2, grady2, tc):
    """trackFeatures.py:67-136 under the reference's own name: one feature at one pyramid level, from the planes the caller hands
    over -- the three patches around (x1, y1), the Newton loop (klt_track_iterate_f32 through trackFeaturesUtils), the window's bounds
    with the Python-3 float half-window (3.5 for 7x7, :88-89, :108), the residue test (:113-121: computeIntensityDifference, numpy's
    f32 pairwise sum), retainTrackers and the status priority (:123-135).  Returns (status, x2, y2).
    KLTTrackFeatures does not come through here: klt_track runs every feature and level in one launch."""
    from . import trackFeaturesUtils as tfu
    width, height = tc.window_width, tc.window_height
    hw, hh = width / 2, height / 2
    nc, nr = img1.shape[1], img1.shape[0]
    one_plus_eps = 1.001
    img1Patch = tfu.extractImagePatchSlow(img1, x1, y1, height, width)
    img1GradxPatch = tfu.extractImagePatchSlow(gradx1, x1, y1, height, width)
    img1GradyPatch = tfu.extractImagePatchSlow(grady1, x1, y1, height, width)
    x2, y2, status, iteration = tfu.trackFeatureIterateCKLT(x2, y2, img1GradxPatch, img1GradyPatch, img1Patch, img2, gradx2, grady2, tc)
    if x2 - hw < 0.0 or nc - (x2 + hw) < one_plus_eps or y2 - hh < 0.0 or nr - (y2 + hh) < one_plus_eps:
        status = kltState.KLT_OOB
    if status == kltState.KLT_TRACKED and tc.max_residue is not None:
        if tc.lighting_insensitive:
            raise Exception("Not implemented")
        workingPatch = np.empty((height, width), np.float32)
        imgdiff = np.zeros(workingPatch.size, np.float32)
        tfu.computeIntensityDifference(img1Patch, img2, x2, y2, workingPatch, imgdiff)
        if np.abs(imgdiff).sum() / (width * height) > tc.max_residue:
            status = kltState.KLT_LARGE_RESIDUE
    if tc.retainTrackers:
        return kltState.KLT_TRACKED, x2, y2
    if status in (kltState.KLT_SMALL_DET, kltState.KLT_OOB, kltState.KLT_LARGE_RESIDUE):
        return status, x2, y2
    if iteration >= tc.max_iterations:
        return kltState.KLT_MAX_ITERATIONS, x2, y2
    return kltState.KLT_TRACKED, x2, y2


def _outOfBounds(x, y, ncols, nrows, borderx, bordery):
    """trackFeatures.py:140-141"""
    return x < borderx or x > ncols - 1 - borderx or y < bordery or y > nrows - 1 - bordery


def _prepare_pair(tc, ctx, img1, img2, speculate=False):
    """ComputeImagePyramids (trackFeatures.py:146-196) on the device: both frames in their slots, their pyramids built or being
    built.  Returns (slot1, slot2, ncols, nrows, doubts).

    `speculate`: a slot that passes the fast rejects (size, mode, 1024-pixel lattice) is TAKEN as holding the image, and listed in
    `doubts` as (slot, key), frame 2 first: the caller enqueues its device work at once, compares every byte while the device
    runs (`FrameCache.verify`), and for a frame that turns out to differ calls `_resend` and enqueues its work again.  Without it
    every byte is compared before anything is enqueued and `doubts` is empty."""
    ncols, nrows = _image_size(img1)
    assert _image_size(img2) == (ncols, nrows)
    _fix_window(tc)
    frames = cache_of(tc)
    if frames.handles and not ctx.configured_for(tc):
        frames.keep_all_handles()                     # new parameters void every pyramid of the context: kept handles fetch theirs first
    ctx.configure(tc)
    s1, s2, _ = _slots_of(tc)
    doubts = []
    sure = frames.trusting() or not speculate

    def locate(key, slots):
        """(slot that holds the image -- or, speculating, probably does --, True when that is still to be verified)"""
        if not sure and len(slots) > 1:
            # speculating: the slot that was filled from this very object goes first (an image edited in place matches NEITHER slot's
            # pixels, but two different frames of a static scene can share a lattice: taking the other one's slot costs a wasted
            # tracker launch and a resend when the slot filled from this object holds the exact image -- ADVICE r5)
            slots = sorted(slots, key=lambda s: not frames.filled_from(key, s))
        for s in slots:
            if frames.plausible(key, s, ctx):
                if not sure:
                    return s, True
                if frames.verify(key, s):
                    return s, False
        return None, False

    resident = tc.sequentialMode and tc.pyramid_last is not None
    if resident and not ctx.pyramids_valid(s1):
        # the pyramid geometry / taps changed since the last call (or another tracking context rebuilt the slot):
        # the reference would go on with the stale pyramid_last; rebuilding frame 1 from img1 is the useful reading
        resident = False
    if resident:
        if tc.pyramid_last.ncols[0] != ncols or tc.pyramid_last.nrows[0] != nrows:
            from .error import KLTError
            KLTError("(KLTTrackFeatures) Size of incoming image ({0} by {1}) is different from size of previous image "
                     "({2} by {3})".format(ncols, nrows, tc.pyramid_last.ncols[0], tc.pyramid_last.nrows[0]))
        k2 = FrameKey(img2)
        at2, doubt2 = locate(k2, (s2,))
        if at2 is None:
            frames.send(ctx, s2, k2)
            ctx.build_pyramids(s2, sync=False)
        else:
            if doubt2:
                doubts.append((s2, k2))
            if not ctx.pyramids_valid(s2):
                frames.keep_handles(ctx, s2)
                ctx.build_pyramids(s2, sync=False)
    else:
        # A slot that already holds exactly the pixels of one of the two images (every byte compared -- _frames.py) keeps it: the
        # reference converts and rebuilds both on every call, example1's ping-pong (example1.py:53-56) the same two 200 times.
        k1, k2 = FrameKey(img1), FrameKey(img2)
        (at1, doubt1), (at2, doubt2) = locate(k1, (s1, s2)), locate(k2, (s1, s2))
        if at1 == s2 or at2 == s1:                    # the pair arrives the other way round (or shifted by one frame)
            ctx.swap_slots(s1, s2)
            frames.swap(s1, s2)
            at1 = {s1: s2, s2: s1}.get(at1)
            at2 = {s1: s2, s2: s1}.get(at2)
        build = []
        for slot, key, at, doubt in ((s2, k2, at2, doubt2), (s1, k1, at1, doubt1)):      # frame 2 first: in a video it is the new one
            if at != slot:
                frames.send(ctx, slot, key)
                build.append(slot)
            else:
                if doubt:
                    doubts.append((slot, key))
                if not ctx.pyramids_valid(slot):      # the frame is there, the pyramid geometry / taps changed since
                    frames.keep_handles(ctx, slot)
                    build.append(slot)
        if build:
            ctx.build_pyramids_batch(build)           # frames share every kernel launch
    return s1, s2, ncols, nrows, doubts


def _prepared_scores_pay(tc, ncols, nrows):
    """the replacement selection uses scores prepared ahead only behind its candidate cut: more than 262 144 candidates, a minimum
    distance to enforce, every pixel a candidate (klt_select_begin_async)"""
    bx, by = max(tc.borderx, tc.window_width / 2.0), max(tc.bordery, tc.window_height / 2.0)
    return (tc.mindist > 0 and tc.nSkippedPixels == 0 and tc.smoothBeforeSelecting
            and (ncols - 2 * int(bx)) * (nrows - 2 * int(by)) > 262144)


def _resend(tc, ctx, slot, key):
    """a frame taken as resident on the strength of its lattice differs from what the slot holds: send it and rebuild"""
    cache_of(tc).send(ctx, slot, key)
    ctx.build_pyramids(slot, sync=False)


def ComputeImagePyramids(tc, img1, img2):
    """trackFeatures.py:146-196: (pyramid1, pyramid1_gradx, pyramid1_grady, pyramid2, pyramid2_gradx, pyramid2_grady) of the two
    images -- smoothed with sigma = smooth_sigma_fact * max(w, h), reduced nPyramidLevels - 1 times, differentiated per level.  In
    sequential mode with pyramids kept from the last KLTTrackFeatures call the first three are tc.pyramid_last* and img1 is not looked
    at (:152-161).  The pyramids are built and stay on the device (one launch sequence for both images); the objects returned have the
    reference's KLTPyramid attributes, and `pyramid.img[level]` is the float32 plane, downloaded when first looked at."""
    ctx = context_of(tc)
    with ctx.lock:
        ctx.settle_deferred()
        return _compute_image_pyramids(tc, ctx, img1, img2)


def _compute_image_pyramids(tc, ctx, img1, img2):
    s1, s2, ncols, nrows, _ = _prepare_pair(tc, ctx, img1, img2)
    # (no settle_frames here: nothing has been waited for, the staged frames may still be on their way -- the next call that wants
    # their pinned buffers waits for the copies)
    if tc.sequentialMode and tc.pyramid_last is not None and getattr(tc.pyramid_last, "_gen", None) == ctx.slot_generation(s1):
        first = (tc.pyramid_last, tc.pyramid_last_gradx, tc.pyramid_last_grady)
    else:
        first = _pyramid_handles(tc, ctx, s1, ncols, nrows)
    if tc.writeInternalImages:
        from .klt_util import KLTWriteFloatImageToPGM
        second = _pyramid_handles(tc, ctx, s2, ncols, nrows)
        for i in range(tc.nPyramidLevels):                   # trackFeatures.py:186-194
            for tag, (p, gx, gy) in (("i", first), ("j", second)):
                KLTWriteFloatImageToPGM(p.img[i], "kltimg_tf_{0}{1}.pgm".format(tag, i))
                KLTWriteFloatImageToPGM(gx.img[i], "kltimg_tf_{0}{1}_gx.pgm".format(tag, i))
                KLTWriteFloatImageToPGM(gy.img[i], "kltimg_tf_{0}{1}_gy.pgm".format(tag, i))
        return first + second
    return first + _pyramid_handles(tc, ctx, s2, ncols, nrows)


def KLTTrackFeatures(tc, img1, img2, featurelist):
    """trackFeatures.py:205-409 (translation model; the affine branch :347-399 calls functions the
    reference never defines)."""
    if KLT_verbose >= 1:
        ncols, nrows = _image_size(img1)
        print("(KLT) Tracking {0} features in a {1} by {2} image...  ".format(
            KLTCountRemainingFeatures(featurelist), ncols, nrows))
    ctx = context_of(tc)
    with ctx.lock:                                    # one KLT* call at a time per device context (backend.default_context)
        ctx.settle_deferred()
        _track_locked(ctx, tc, img1, img2, featurelist)
    if KLT_verbose >= 1:
        print("\n\t{0} features successfully tracked.".format(KLTCountRemainingFeatures(featurelist)))


def _track_locked(ctx, tc, img1, img2, featurelist):
    affine = tc.affineConsistencyCheck >= 0
    # The affine check updates its per-feature state in place, so its launch cannot be repeated: nothing is taken on trust there.
    s1, s2, ncols, nrows, doubts = _prepare_pair(tc, ctx, img1, img2, speculate=not affine)

    nfeat = len(featurelist)
    store = shared_store(featurelist)
    fl_in = features_to_array(featurelist, ctx.host_records(nfeat)[0], store)      # pinned: goes up without a staging copy
    state = None
    if affine:
        # The reference calls _am_trackFeatureAffine here but never defines it (trackFeatures.py:347-399); behaviour
        # follows upstream KLT 1.3.4.  The per-feature templates / A matrices live on the device, keyed
        # by the feature list object.
        state = _affine_state_of(tc, ctx, featurelist)
    ctx.track_enqueue(s1, s2, nfeat, state)
    # The device is tracking; now every byte of the frames that were taken as resident is compared with what their slots were
    # filled from.  One that differs (the same array edited in place off the lattice, ...) is sent and built now and the tracker
    # runs again on the same input records: the records that come back are those of the last launch.
    again = False
    for slot, key in doubts:
        if not cache_of(tc).verify(key, slot):
            _resend(tc, ctx, slot, key)
            again = True
    if again:
        ctx.track_enqueue(s1, s2, nfeat, state, upload=False)
    # A tracking context whose last KLTTrackFeatures call was followed by KLTReplaceLostFeatures (the loop of a video script): the
    # list-independent half of that replacement -- summed-area tables and eigenvalue keys of frame 2's level 0, 50 us at 1080p -- is
    # enqueued now, behind the tracker, and runs while the host moves the columns and finds its way into the replacement call
    # (klt_select_prepare_async; the score set follows the slot swap below and dies unused with the slot's next build).
    marked = False
    # (same box, alternating: 407 / 418 / 420 us per frame without, 386 / 365 / 387 with)
    if tc.sequentialMode and not affine and tc.__dict__.pop("_klt_replaced_after_track", False) and _prepared_scores_pay(tc, ncols, nrows):
        marked = ctx.track_mark()
        if marked:
            try:
                ctx.select_prepare(s2)
            except KltBackendError:                         # (an optimisation only: the replacement scores the frame itself then)
                pass
    fl_out = ctx.track_complete(nfeat, marked=marked)
    if affine:
        rec = ctx.affine_download(state, nfeat)
    if store is not None:
        # whole columns at once (the reference walks the list: trackFeatures.py:288-399).  The records are final for every row:
        # the tracked position, (-1, -1, status) for a feature that was lost, the input record for one that was not live (:253).
        dead = fl_in["val"] < 0
        live = ~dead
        np.copyto(store.x, fl_out["x"], where=live)
        np.copyto(store.y, fl_out["y"], where=live)
        np.copyto(store.val, fl_out["val"], where=live)
        np.logical_and(store.xint, dead, out=store.xint)      # tracked or lost: Python floats from here on
        np.logical_and(store.yint, dead, out=store.yint)
        store.changed()
        if affine:
            ok = live & (fl_out["val"] == kltState.KLT_TRACKED)
            cols = store.aff_columns()
            for name, key in (("aff_x", "aff_x"), ("aff_y", "aff_y"), ("aff_Axx", "Axx"), ("aff_Ayx", "Ayx"), ("aff_Axy", "Axy"),
                              ("aff_Ayy", "Ayy")):
                cols[name][live] = rec[key][live]
            has_tpl = live & (rec["valid"] != 0) & ok
            for col in store.aff_img.values():
                col[live] = None
                col[has_tpl] = _DEVICE_TEMPLATE
        elif store.aff is not None:
            lost = live & (fl_out["val"] != kltState.KLT_TRACKED)
            for col in store.aff_img.values():
                col[lost] = None
    else:
        olds = fl_in["val"].tolist()
        xs, ys, vals = fl_out["x"].tolist(), fl_out["y"].tolist(), fl_out["val"].tolist()
        if affine:
            rcols = [rec[k].tolist() for k in ("aff_x", "aff_y", "Axx", "Ayx", "Axy", "Ayy", "valid")]
        for i, feat in enumerate(featurelist):
            if olds[i] < 0:
                continue                                  # only live features are tracked (:253)
            if affine:
                feat.aff_x, feat.aff_y = rcols[0][i], rcols[1][i]
                feat.aff_Axx, feat.aff_Ayx, feat.aff_Axy, feat.aff_Ayy = rcols[2][i], rcols[3][i], rcols[4][i], rcols[5][i]
                feat.aff_img = feat.aff_img_gradx = feat.aff_img_grady = (_DEVICE_TEMPLATE if rcols[6][i] else None)
            if vals[i] == kltState.KLT_TRACKED:
                feat.x = xs[i]
                feat.y = ys[i]
                feat.val = kltState.KLT_TRACKED
            else:
                feat.x = -1.0
                feat.y = -1.0
                feat.val = vals[i]
                feat.aff_img = feat.aff_img_gradx = feat.aff_img_grady = None

    settle_frames(ctx)                # (the results are back: the staged frames left their buffers long ago)
    if tc.sequentialMode:
        ctx.swap_slots(s1, s2)                        # frame-2 pyramids become frame 1 (:401-404)
        cache_of(tc).swap(s1, s2)
        tc.pyramid_last, tc.pyramid_last_gradx, tc.pyramid_last_grady = _pyramid_handles(tc, ctx, s1, ncols, nrows)
