"""Public KLT types: tracking context, feature record, status codes.

Reference: klt.py.  Same names, same attribute set and the same derived values *as the
reference computes them under Python 3* (true division: a 7x7 window gives a half-width of
3.5 and a default border of 30.0, not the 24 of the C library -- SURVEY.md A.1).  The
numeric state lives in a `klt_params` POD (include/klt_gpu.h) pushed across the C ABI.
"""
from __future__ import print_function

import gc as _gc
import math
import os as _os
import sys as _sys
import threading as _threading
import time
import weakref as _weakref
from itertools import repeat as _repeat

import numpy as np

from .convolve import KLTGetKernelWidths
from .error import KLTError, KLTWarning  # noqa: F401  (re-exported like the reference's star imports)
from .klt_util import KLTComputeSmoothSigma

# example1.py:52 uses time.clock(), removed in Python 3.8; give it back so that the reference's
# own driver script runs unchanged on top of this backend.
if not hasattr(time, "clock"):
    time.clock = time.perf_counter


class kltState:
    """klt.py:23-29"""
    KLT_TRACKED = 0
    KLT_NOT_FOUND = -1
    KLT_SMALL_DET = -2
    KLT_MAX_ITERATIONS = -3
    KLT_OOB = -4
    KLT_LARGE_RESIDUE = -5


def _pyramidSigma(tc):
    """klt.py:196-197"""
    return tc.pyramid_sigma_fact * tc.subsampling


class KLT_TrackingContext:
    """klt.py:42-190.  Plain attributes; nothing is validated until a KLT* call uses them."""

    def __init__(self):
        self.mindist = 10
        self.window_width = 7
        self.window_height = 7
        self.sequentialMode = False
        self.retainTrackers = False
        self.smoothBeforeSelecting = True
        self.writeInternalImages = False
        self.lighting_insensitive = False
        self.min_eigenvalue = 1
        self.min_determinant = 0.01
        self.max_iterations = 10
        self.min_displacement = 0.1
        self.max_residue = None
        self.grad_sigma = 1.0
        self.smooth_sigma_fact = 0.1
        self.pyramid_sigma_fact = 0.9
        self.step_factor = 1.0
        self.nSkippedPixels = 0
        self.pyramid_last = None
        self.pyramid_last_gradx = None
        self.pyramid_last_grady = None
        # affine consistency check (klt.py:67-73).  Not implemented by the reference.
        self.affineConsistencyCheck = -1
        self.affine_window_width = 15
        self.affine_window_height = 15
        self.affine_max_iterations = 10
        self.affine_max_residue = 10.0
        self.affine_min_displacement = 0.02
        self.affine_max_displacement_differ = 1.5

        self.KLTChangeTCPyramid(15)
        self.KLTUpdateTCBorder()

    # -- window sanity, shared by every entry point of the reference (klt.py:87-101, :143-157)
    def _check_window(self, who):
        for name, what in (("window_width", "width"), ("window_height", "height")):
            v = getattr(self, name)
            if v % 2 != 1:
                v += 1
                setattr(self, name, v)
                KLTWarning("({0}) Window {1} must be odd.  Changing to {2}.\n".format(who, what, v))
        for name, what in (("window_width", "width"), ("window_height", "height")):
            if getattr(self, name) < 3:
                setattr(self, name, 3)
                KLTWarning("({0}) Window {1} must be at least three.  \nChanging to 3.\n".format(who, what))

    def KLTChangeTCPyramid(self, search_range):
        """Choose nPyramidLevels / subsampling for a search range -- klt.py:84-128."""
        self._check_window("KLTChangeTCPyramid")
        half = min(self.window_width, self.window_height) / 2.0
        ratio = float(search_range) / half
        if ratio < 1.0:
            self.nPyramidLevels = 1
        elif ratio <= 9.0:
            self.nPyramidLevels = 2
            self.subsampling = 2 if ratio <= 3.0 else (4 if ratio <= 5.0 else 8)
        else:
            # search_range = half * (8^L - 1) / 7, rounded up
            self.nPyramidLevels = int(math.log(7.0 * ratio + 1.0) / math.log(8.0) + 0.99)
            self.subsampling = 8

    def KLTUpdateTCBorder(self):
        """Border lost to convolution and windows -- klt.py:137-189 (Python-3 float halves)."""
        self._check_window("KLTUpdateTCBorder")
        levels = self.nPyramidLevels
        ss = self.subsampling
        window_half = max(self.window_width, self.window_height) / 2
        smooth_half = KLTGetKernelWidths(KLTComputeSmoothSigma(self))[0] / 2
        pyramid_half = KLTGetKernelWidths(_pyramidSigma(self))[0] / 2
        invalid = smooth_half
        for _ in range(1, levels):
            invalid = int((float(invalid) + pyramid_half) / ss + 0.99)
        border = (invalid + window_half) * ss ** (levels - 1)
        self.borderx = border
        self.bordery = border


class _StaleColumn:
    """What a store's lx / ly / lv are while the plain-list form of the columns is out of date: indexing it rebuilds the three lists
    (which replace the stand-ins on the store) and answers from the new one."""
    __slots__ = ("_store", "_k")

    def __init__(self, store_ref, k):
        self._store, self._k = store_ref, k

    def __getitem__(self, i):
        return self._store()._lists()[self._k][i]


class _FeatureStore:
    """Column storage behind the KLT_Feature objects of one feature list.

    The reference keeps x / y / val (and the affine fields) as attributes of 5000 separate Python objects and touches every one of
    them on every call (selectGoodFeatures.py:116-128, trackFeatures.py:288-399).  Here the objects are thin views of rows of
    shared numpy columns, so KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures move whole columns to and from the
    16-byte device records instead of looping over features; a feature object reads its row when somebody looks at it.
    x / y are kept as float64 (exact for the reference's values: integers after selection, f32-valued after tracking) plus a flag
    that remembers whether the reference would hold a Python int there (`print` shows 86, not 86.0).
    The store does not point at its feature objects (they point at it): a dropped list is freed by reference counting, not by the
    cycle collector.  `owner` is a weak reference to the KLT_FeatureList that was made with the store."""

    __slots__ = ("x", "y", "val", "xint", "yint", "aff", "aff_img", "owner", "hooks", "lx", "ly", "lv", "_stale", "tagged", "kept", "__weakref__")
    _AFF_DEFAULTS = (("aff_x", -1.0), ("aff_y", -1.0), ("aff_Axx", 1.0), ("aff_Ayx", 0.0), ("aff_Axy", 0.0), ("aff_Ayy", 1.0))

    def __init__(self, n):
        self.x = np.full(n, -1.0)
        self.y = np.full(n, -1.0)
        self.val = np.full(n, kltState.KLT_NOT_FOUND, np.int64)
        self.xint = np.ones(n, bool)
        self.yint = np.ones(n, bool)
        self.aff = None               # {name: float64 column}, created when an affine field is first written
        self.aff_img = None           # {name: object column} for aff_img / aff_img_gradx / aff_img_grady
        self.owner = None             # weakref to the KLT_FeatureList whose `_canon` lists the objects of rows 0 .. n-1
        self.hooks = None             # callbacks for the end of the features' life (device-side affine state: trackFeatures.py)
        self.tagged = False           # somebody gave a feature of this store an attribute of their own (or looked at its __dict__)
        self.kept = None              # a private copy of a caller's plain list found to be exactly this store's rows (`shared_store`)
        # lx / ly / lv: the columns as plain lists of the Python values the features show (ints where the reference holds ints) -- what a
        # feature's x / y / val read, one C-level list index per attribute.  After the columns were written (`changed()`) they are
        # stand-ins (`_StaleColumn`) that rebuild all three lists when first indexed: a loop over every feature of a list pays one
        # conversion per column instead of two numpy scalar reads per attribute, a loop that never looks at a feature pays nothing.
        ref = _weakref.ref(self)
        self._stale = (_StaleColumn(ref, 0), _StaleColumn(ref, 1), _StaleColumn(ref, 2))
        self.lx, self.ly, self.lv = self._stale

    def when_features_die(self, callback):
        """`callback()` runs when the KLT_FeatureList made with this store is dropped (what is keyed by that list -- the device-side
        affine state -- is unreachable from then on, and the objects may be handed to a new list: `_recycled`), at the latest when
        the last feature of the store is gone.  Callbacks must be idempotent.  ONE finalizer per store runs whatever the list
        holds at the end (a finalizer per call would pile up in weakref.finalize's registry for as long as a recycled store
        lives -- one per select + register round, each pinning its callback's closure: ADVICE r5)."""
        if self.hooks is None:
            self.hooks = []
            _weakref.finalize(self, _run_hook_list, self.hooks)      # (holds the LIST, never the store)
        self.hooks.append(callback)

    def changed(self):
        """the columns were written (a KLT* call, a feature's setter): the lists the features read from are stale"""
        self.lx, self.ly, self.lv = self._stale

    def _lists(self):
        def column(values, ints):
            o = values.astype(object)                              # Python floats ...
            if ints.any():
                o[ints] = values[ints].astype(np.int64).astype(object)      # ... and Python ints where the reference holds ints
            return o.tolist()
        self.lx, self.ly, self.lv = ls = (column(self.x, self.xint), column(self.y, self.yint), self.val.tolist())
        return ls

    def _run_hooks(self):
        if self.hooks:
            _run_hook_list(self.hooks)

    def _reset(self):
        """back to n lost features nobody has looked at (the objects of a dropped list serve the next one: `_recycled`)"""
        self._run_hooks()
        self.x.fill(-1.0)
        self.y.fill(-1.0)
        self.val.fill(kltState.KLT_NOT_FOUND)
        self.xint.fill(True)
        self.yint.fill(True)
        self.aff = self.aff_img = self.owner = self.kept = None
        self.changed()

    def __len__(self):
        return self.x.shape[0]

    def __reduce__(self):                   # pickles / deep-copies as its columns (the owner is a weak reference)
        return (_restored_store, (self.x, self.y, self.val, self.xint, self.yint, self.aff, self.aff_img))

    def aff_columns(self):
        if self.aff is None:
            n = len(self)
            self.aff = {name: np.full(n, dflt) for name, dflt in self._AFF_DEFAULTS}
            self.aff_img = {name: np.full(n, None, object) for name in ("aff_img", "aff_img_gradx", "aff_img_grady")}
        return self.aff

    def reset_affine(self, rows):
        """rows (index array / mask) back to the state of newly placed features (selectGoodFeatures.py:120-128)"""
        if self.aff is None:
            return
        for name, dflt in self._AFF_DEFAULTS:
            self.aff[name][rows] = dflt
        for col in self.aff_img.values():
            col[rows] = None


def _run_hook_list(hooks):
    """run and forget the callbacks registered so far (the list object stays the one the store's finalizer holds)"""
    pending = hooks[:]
    del hooks[:]
    for h in pending:
        h()


def _restored_store(x, y, val, xint, yint, aff, aff_img):
    s = _FeatureStore(0)
    s.x, s.y, s.val, s.xint, s.yint, s.aff, s.aff_img = x, y, val, xint, yint, aff, aff_img
    s.changed()
    return s


_item = list.__getitem__            # the [store, row] pair of a feature: `feat[0]` itself is hidden from callers
_pair_len = list.__len__


def _own(feat):
    """the store of a feature; one made on its own (`KLT_Feature()`, as the reference's scripts write) gets its one-row store here, when
    it is first looked at (a Python-level __new__ / __init__ would make every object of a 5000-feature list pay for a Python call)"""
    if not _pair_len(feat):
        list.extend(feat, (_FeatureStore(1), 0))
    return _item(feat, 0)


def _coord_setter(col, flag):
    def put(self, value):
        s, i = _own(self), _item(self, 1)
        getattr(s, col)[i] = value
        getattr(s, flag)[i] = isinstance(value, (int, np.integer)) and not isinstance(value, bool)
        s.changed()
    return put


def _aff_property(name, dflt):
    def get(self):
        s = _own(self)
        return dflt if s.aff is None else float(s.aff[name][_item(self, 1)])

    def put(self, value):
        _own(self).aff_columns()[name][_item(self, 1)] = value
    return property(get, put)


def _aff_img_property(name):
    def get(self):
        s = _own(self)
        return None if s.aff is None else s.aff_img[name][_item(self, 1)]

    def put(self, value):
        s = _own(self)
        s.aff_columns()
        s.aff_img[name][_item(self, 1)] = value
    return property(get, put)


class _FeatureObject(list):
    """A list subclass WITHOUT __slots__: its instances carry a lazily created __dict__ (nothing is allocated until somebody sets an
    attribute) and can be weakly referenced.  A class of its own so that KLT_Feature can put a property in front of the instance
    dictionary (`_instance_dict`)."""


_instance_dict = _FeatureObject.__dict__["__dict__"]        # the getset descriptor that hands out / replaces an instance's dictionary
_FIELDS = frozenset(("x", "y", "val", "aff_x", "aff_y", "aff_Axx", "aff_Ayx", "aff_Axy", "aff_Ayy", "aff_img", "aff_img_gradx", "aff_img_grady"))
_generic_setattr = object.__setattr__


def _no_sequence(what):
    def refuse(self, *args, **kw):
        raise TypeError("'KLT_Feature' object %s" % what)
    return refuse


class KLT_Feature(_FeatureObject):
    """klt.py:249-263.  The reference's __init__ assigns locals only; real attributes are set on first placement
    (selectGoodFeatures.py:117-128) -- a KLT_Feature there is an attribute bag.  Here x, y, val and the affine-consistency fields
    always exist and read row `_i` of a _FeatureStore `_s` (its own one-row store when created on its own, the list's shared store
    when it comes from KLTSelectGoodFeatures / KLTCreateFeatureList); ANY OTHER attribute is the caller's (`feat.track_id = 7`) and
    lives in the object's own dictionary, as on the reference's plain objects; a feature can be weakly referenced as they can.

    Underneath it is the PAIR [store, row] -- a `list` subclass with neither __new__ nor __init__ of its own -- because that is the
    cheapest object with a dictionary and weak references that CPython can make in bulk: a 5000-feature list is
    `map(KLT_Feature, zip(...))`, the type call all in C (0.22 ms here; a tuple subclass, rounds 5 / 6a: 0.27 ms and no weak references;
    5000 calls of a Python `__init__`: 0.57 ms), and the list KLTSelectGoodFeatures hands out is a complete list of feature objects
    as the reference's is.  Nothing of the pair shows: a feature has no length, is not iterable or subscriptable, has none of a list's
    methods, is always true, equals and hashes by identity (numpy's array constructor makes a 1-D object array of a list of features;
    `feat in some_list` asks for this very object), pickles and deep-copies as a feature of its own (values, affine fields, own
    attributes -- not the list's column store).  The objects of a dropped list serve the next one (`_recycled`) only when no
    feature of the list was ever given an attribute of its own and none is weakly referenced."""

    _s = property(_own)
    _i = property(lambda self: _item(self, 1) if _own(self) is not None else 0)

    @property
    def x(self):
        try:
            return _item(self, 0).lx[_item(self, 1)]
        except IndexError:                      # a feature made on its own, looked at for the first time (anything else is an error)
            if _pair_len(self):
                raise
            return _own(self).lx[0]

    @property
    def y(self):
        try:
            return _item(self, 0).ly[_item(self, 1)]
        except IndexError:
            if _pair_len(self):
                raise
            return _own(self).ly[0]

    @property
    def val(self):
        try:
            return _item(self, 0).lv[_item(self, 1)]
        except IndexError:
            if _pair_len(self):
                raise
            return _own(self).lv[0]

    x = x.setter(_coord_setter("x", "xint"))
    y = y.setter(_coord_setter("y", "yint"))

    @val.setter
    def val(self, value):
        s = _own(self)
        s.val[_item(self, 1)] = value
        s.changed()

    aff_x = _aff_property("aff_x", -1.0)
    aff_y = _aff_property("aff_y", -1.0)
    aff_Axx = _aff_property("aff_Axx", 1.0)
    aff_Ayx = _aff_property("aff_Ayx", 0.0)
    aff_Axy = _aff_property("aff_Axy", 0.0)
    aff_Ayy = _aff_property("aff_Ayy", 1.0)
    aff_img = _aff_img_property("aff_img")
    aff_img_gradx = _aff_img_property("aff_img_gradx")
    aff_img_grady = _aff_img_property("aff_img_grady")

    # ---- the attribute bag.  Own attributes go to the instance dictionary as on any object; the store remembers that one of its
    # features carries something of the caller's (a store like that is never recycled: a recycled object starts clean).
    def __setattr__(self, name, value):
        if name not in _FIELDS:
            _own(self).tagged = True
        _generic_setattr(self, name, value)

    @property
    def __dict__(self):
        _own(self).tagged = True                # (vars(feat)[...] = ... writes without __setattr__)
        return _instance_dict.__get__(self)

    @__dict__.setter
    def __dict__(self, value):
        _own(self).tagged = True
        _instance_dict.__set__(self, value)

    # ---- nothing of the pair shows
    __len__ = _no_sequence("has no len()")
    __iter__ = __reversed__ = _no_sequence("is not iterable")
    __getitem__ = __setitem__ = __delitem__ = _no_sequence("is not subscriptable")
    __contains__ = _no_sequence("is not a container")
    __add__ = __iadd__ = __mul__ = __imul__ = __rmul__ = _no_sequence("is not a sequence")
    __lt__ = __le__ = __gt__ = __ge__ = lambda self, other: NotImplemented
    __eq__ = object.__eq__
    __ne__ = object.__ne__
    __hash__ = object.__hash__
    __class_getitem__ = None

    def _no_list_method(self):
        raise AttributeError("'KLT_Feature' object has no such attribute (it is not a list)")
    append = extend = insert = pop = remove = clear = index = count = sort = reverse = copy = property(_no_list_method)
    del _no_list_method

    def __bool__(self):
        return True

    def __reduce_ex__(self, protocol):
        # a feature of its own: the values it shows (ints where it shows ints), the affine fields if any were written, own attributes
        s, i = _own(self), _item(self, 1)
        aff = None if s.aff is None else ({k: float(v[i]) for k, v in s.aff.items()}, {k: v[i] for k, v in s.aff_img.items()})
        return (_restored_feature, (self.x, self.y, self.val, aff, dict(_instance_dict.__get__(self)) if s.tagged else None))

    def _reset_affine(self):
        """Back to the state of a newly placed feature (selectGoodFeatures.py:120-128)."""
        _own(self).reset_affine(_item(self, 1))

    def __repr__(self):
        return "<KLT_Feature x={0!r} y={1!r} val={2!r}>".format(self.x, self.y, self.val)


def _restored_feature(x, y, val, aff, own):
    f = KLT_Feature()
    f.x, f.y, f.val = x, y, val
    if aff is not None:
        for k, v in aff[0].items():
            setattr(f, k, v)
        for k, v in aff[1].items():
            setattr(f, k, v)
    if own:
        f.__dict__.update(own)
    return f


def _row_features(store, rows):
    """the feature objects of rows `rows` of `store`, made in one C-level pass (the type call of a list subclass without __new__ /
    __init__ of its own: list's constructor fills the object from the (store, row) pair)"""
    return map(KLT_Feature, zip(_repeat(store), rows))


class KLT_FeatureList(list):
    """The list KLTSelectGoodFeatures / KLTCreateFeatureList hand out (the reference builds `[KLT_Feature() for i in
    range(nFeatures)]`, selectGoodFeatures.py:143; its own KLT_FeatureList is commented out, klt.py:266-270).  It IS a list of
    KLT_Feature objects -- complete when it is handed out, as the reference's -- made in one C-level pass with the cycle collector
    paused (`_fill`), which KLTSelectGoodFeatures runs while the device is still selecting.  A tracking loop that only hands the
    list from one KLT* call to the next never looks at one of the objects: the calls work on the column store.
    In the LAZY mode (opt-in: `klt.LAZY_FEATURE_LISTS = True`, or KLT_LAZY_FEATURE_LISTS=1 in the environment) the objects are
    made when somebody first touches an element (indexing, iteration, any list method); `len()`, truth value and the KLT* calls
    themselves do not.  The price of that mode: C code that reads the list's storage directly (`PySequence_Fast` users such as
    slice assignment `a[0:0] = fl`, numpy's array constructor, Cython functions with list-typed arguments) sees an empty list
    until something has touched it."""

    __slots__ = ("_store", "_pending", "_canon", "__weakref__")

    def __init__(self, store):
        list.__init__(self)
        self._store = store
        self._pending = len(store)
        self._canon = None            # private copy of the complete list: what `shared_store` compares a caller's list with

    def _fill(self):
        n = self._pending
        if n:
            self._pending = 0
            store = self._store
            # 5000 new container objects cross the collector's young-generation threshold seven times; the young objects are all
            # reachable, so those passes find nothing and double the cost of the fill.  Paused for the length of one C call; a
            # collector the caller has disabled stays disabled.
            paused = _gc.isenabled()
            if paused:
                _gc.disable()
            try:
                list.extend(self, _row_features(store, range(n)))
            finally:
                if paused:
                    _gc.enable()
            self._canon = list.copy(self)       # the caller's list may be edited
            store.owner = _weakref.ref(self)

    def __len__(self):
        return self._pending or list.__len__(self)

    def __del__(self):
        # The caller's list is gone.  Its feature objects (kept alive by the private copy) are offered to the next list of the same
        # length: whether somebody still holds one of them is looked at when they are about to be used again (`_recycled`).
        try:
            canon = self._canon
            self._store._run_hooks()
            if canon is not None and RECYCLE_FEATURE_OBJECTS:
                _offer(self._store, canon)
        except Exception:                           # noqa: BLE001 -- interpreter shutdown
            pass

    def __radd__(self, other):
        # `[] + fl`, `sum([fl1, fl2], [])`: list.__add__ reads the right operand's storage directly, so the reflected method (tried
        # first for a subclass on the right) fills the list before anything is concatenated
        self._fill()
        return list(other) + list(self)

    def __reduce_ex__(self, protocol):
        self._fill()
        return (list, (list(self),))            # pickles (and copies) as the plain list of features it stands for


def _filled_first(name):
    inherited = getattr(list, name)

    def method(self, *args, **kw):
        self._fill()
        return inherited(self, *args, **kw)
    method.__name__ = name
    method.__doc__ = inherited.__doc__
    return method


for _name in ("__getitem__", "__setitem__", "__delitem__", "__iter__", "__reversed__", "__contains__", "__add__", "__iadd__", "__mul__",
              "__imul__", "__rmul__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__repr__", "append", "extend", "insert", "pop",
              "remove", "clear", "index", "count", "sort", "reverse", "copy"):
    setattr(KLT_FeatureList, _name, _filled_first(_name))
KLT_FeatureList.__hash__ = None


LAZY_FEATURE_LISTS = _os.environ.get("KLT_LAZY_FEATURE_LISTS") == "1"

# ---- the feature objects of a dropped list serve the next list of the same length -------------------------------------------
# A loop such as `fl = KLTSelectGoodFeatures(tc, img, 5000)` per frame drops a complete list of 5000 objects per call and makes
# 5000 new ones: 0.3 ms to create, 0.1 ms for the cycle collector's look at them, 0.07 ms to free -- three times what the device
# needs to select the features (profiles/README.md).  The objects of a dropped list are therefore kept (with their store) and,
# provided NOBODY holds one of them any more -- every reference count is looked at, 0.1 ms --, handed out again as the next
# list: same objects, store reset to n lost features, callbacks registered with `when_features_die` run.  A list one of whose
# features was given an attribute of the caller's own (`store.tagged`) is never handed out again: a recycled object starts clean.
# CPython with the GIL only (sys.getrefcount is exact there; a free-threaded build defers and biases reference counts, so a feature
# still in use could look unheld: off when sys._is_gil_enabled() says so); KLT_NO_FEATURE_RECYCLING=1 in the environment (or
# klt.RECYCLE_FEATURE_OBJECTS = False) turns it off.
_getrefcount = getattr(_sys, "getrefcount", None)
RECYCLE_FEATURE_OBJECTS = (_getrefcount is not None and getattr(_sys, "_is_gil_enabled", lambda: True)()
                           and _os.environ.get("KLT_NO_FEATURE_RECYCLING") != "1")
_POOL_LENGTHS, _POOL_DEPTH = 4, 2
_pool = {}                      # list length -> [(store, canon), ...] (newest last)
_pool_lock = _threading.RLock()         # re-entrant: the collector may finalize another dropped list while `_offer` holds it


def _offer(store, canon):
    if store.tagged:
        return
    with _pool_lock:
        stack = _pool.pop(len(canon), None)
        if stack is None:
            stack = []
            while len(_pool) >= _POOL_LENGTHS:
                del _pool[next(iter(_pool))]
        stack.append((store, canon))
        del stack[:-_POOL_DEPTH]
        _pool[len(canon)] = stack                   # (re-inserted last: the dict's order is the order of last use)


def _counts(entry):
    """(highest reference count among the feature objects, the store's count beyond one per feature) of a pool entry that only
    the caller holds -- compared with what the same call gives for an entry made on the spot (`_UNSHARED`)"""
    store, canon = entry
    return max(map(_getrefcount, canon)), _getrefcount(store) - len(canon), max(map(_weakref.getweakrefcount, canon))


def _probe_counts():
    st = _FeatureStore(3)
    entry = (st, list(_row_features(st, range(3))))
    del st
    return _counts(entry)


_UNSHARED = _probe_counts() if _getrefcount is not None else None


def _recycled(n):
    """A filled KLT_FeatureList of n lost features made of the objects of a dropped list, or None."""
    if not RECYCLE_FEATURE_OBJECTS or n == 0:
        return None
    with _pool_lock:
        stack = _pool.get(n)
        entry = stack.pop() if stack else None
    if entry is None or entry[0].tagged or _counts(entry) != _UNSHARED:
        return None                                 # (an entry somebody still holds a feature of is dropped: freed when they let go)
    store, canon = entry
    del entry
    store._reset()
    fl = KLT_FeatureList(store)
    fl._pending = 0
    list.extend(fl, canon)
    fl._canon = canon
    store.owner = _weakref.ref(fl)
    return fl


def new_feature_list(n, fill=None):
    """n lost features sharing one column store (what KLTSelectGoodFeatures / KLTCreateFeatureList hand out).  `fill=False`: the
    caller fills the list itself (KLTSelectGoodFeatures: after the selection has been enqueued)."""
    if (not LAZY_FEATURE_LISTS) if fill is None else fill:
        fl = _recycled(n)
        if fl is None:
            fl = KLT_FeatureList(_FeatureStore(n))
            fl._fill()
        return fl
    return KLT_FeatureList(_FeatureStore(n))


_list_eq = list.__eq__


def shared_store(featurelist):
    """The _FeatureStore whose rows 0 .. n-1 are exactly this list's features, in order -- or None (a list assembled by hand,
    re-ordered, or mixing features of several lists), in which case callers fall back to per-feature access.  The test is one
    C-level list comparison (identity of every element) with the private copy made when the list was filled.  A plain-list copy
    (`fl[:]`, `list(fl)`) of a list this package handed out is recognised while the original is alive by the same comparison, and
    after the original was dropped by looking at every element once (it is row i of the store of element 0: one C-level pass,
    ~0.1 ms at 5000) -- the store then keeps a private copy of that list to compare with on the next call (a reference cycle the
    collector frees with the features; the price of this rare path, not of the usual one)."""
    if type(featurelist) is KLT_FeatureList:
        if featurelist._pending:
            return featurelist._store           # nobody has looked at an element yet: the list is the store's rows by construction
        canon = featurelist._canon
        if canon is not None and _list_eq(featurelist, canon) is True:
            return featurelist._store
    try:
        s = featurelist[0]._s
        owner = s.owner() if s.owner is not None else None
    except (IndexError, AttributeError, TypeError):
        return None
    plain = list(featurelist) if type(featurelist) is not list else featurelist
    if owner is not None and owner._canon is not None:
        return s if len(plain) == len(owner._canon) and _list_eq(plain, owner._canon) is True else None
    if len(plain) != len(s):
        return None
    if s.kept is not None and _list_eq(plain, s.kept) is True:
        return s
    # the list the store was made with is gone: is this list its rows 0 .. n-1, in order, all of them KLT_Feature objects?
    if set(map(type, plain)) != {KLT_Feature} or not all(map(_list_eq, plain, map(list, zip(_repeat(s), range(len(plain)))))):
        return None
    s.kept = plain[:]
    return s


_REC_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("val", np.int32), ("aux", np.int32)])   # == klt_feat


def _lost_records(shape):
    rec = np.zeros(shape, _REC_DTYPE)
    rec["x"] = -1
    rec["y"] = -1
    rec["val"] = kltState.KLT_NOT_FOUND
    return rec


class KLT_FeatureHistory:
    """One feature over all frames (klt.py:272-276 is an empty stub; upstream KLT 1.3.4's KLT_FeatureHistoryRec).
    `rec` is the [nFrames] record array (x, y, val); fh[frame] gives a KLT_Feature copy."""

    def __init__(self, nFrames=0):
        self.nFrames = nFrames
        self.rec = _lost_records(nFrames)

    def __len__(self):
        return self.nFrames

    def __getitem__(self, frame):
        return _feature_of(self.rec[frame])


class KLT_FeatureTable:
    """All features over all frames (klt.py:278-283 is an empty stub; upstream KLT 1.3.4's KLT_FeatureTableRec).
    `rec` is a [nFrames, nFeatures] array of 16-byte records (x, y, val, aux) -- exactly the layout of the
    device-side table KLTTrackSequence fills, so a tracked sequence comes back with ONE download.  Upstream indexes
    feature-major (ft->feature[feat][frame]); `ft.feature(feat, frame)` mirrors that."""

    def __init__(self, nFrames=0, nFeatures=0, _fill=True):
        self.nFrames, self.nFeatures = nFrames, nFeatures
        # (_fill=False: the caller overwrites every row -- KLTTrackSequence's one download -- and marking 1.3 M records lost first would
        # cost as much as tracking a dozen frames)
        self.rec = _lost_records((nFrames, nFeatures)) if _fill else np.empty((nFrames, nFeatures), _REC_DTYPE)

    x = property(lambda self: self.rec["x"])
    y = property(lambda self: self.rec["y"])
    val = property(lambda self: self.rec["val"])

    def feature(self, feat, frame):
        return _feature_of(self.rec[frame, feat])


def _feature_of(r):
    f = KLT_Feature()
    f.x, f.y, f.val = float(r["x"]), float(r["y"]), int(r["val"])
    return f


def KLTPrintTrackingContext(tc):
    """klt.py:285-313 -- same lines, same order."""
    print(tc)
    print("\n\nTracking context:\n")
    for name in ("mindist", "window_width", "window_height", "sequentialMode", "smoothBeforeSelecting",
                 "writeInternalImages"):
        print("\t{0} = {1}".format(name, getattr(tc, name)))
    for name in ("min_eigenvalue", "min_determinant", "min_displacement", "max_iterations", "max_residue",
                 "grad_sigma", "smooth_sigma_fact", "pyramid_sigma_fact", "nSkippedPixels", "borderx", "bordery",
                 "nPyramidLevels", "subsampling"):
        print("\t{0} = {1}".format(name, getattr(tc, name)))
    print("\n\tpyramid_last = {0}".format(tc.pyramid_last))
    print("\tpyramid_last_gradx = {0}".format(tc.pyramid_last_gradx))
    print("\tpyramid_last_grady = {0}".format(tc.pyramid_last_grady))
    print("\n")


def KLTCountRemainingFeatures(fl):
    """klt.py:319-325"""
    s = shared_store(fl)
    if s is not None:
        return int(np.count_nonzero(s.val >= 0))
    return sum(1 for feat in fl if feat.val >= 0)
