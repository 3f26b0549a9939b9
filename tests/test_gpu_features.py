"""KLT_Feature / KLT_FeatureList on the API path (reference klt.py:249-263, selectGoodFeatures.py:117-128, :143): the
reference's attributes with the reference's Python types, own attributes that survive the KLT* calls, the objects of a dropped list
serving the next selection.  (Folded by component from the round-5 file in round 6: the tests are unchanged.)"""
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
