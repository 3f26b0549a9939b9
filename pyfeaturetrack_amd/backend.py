"""Thin object wrapper over the C ABI (include/klt_gpu.h) -- one `Context` per device.

Everything numeric happens behind the ABI in hand-written HIP kernels; this module only
marshals numpy buffers and keeps the tracking-context parameters in sync.  There is no
CPU fallback: creating a Context without a usable MI355X raises KltBackendError.
"""
import collections
import ctypes as C
import os as _os
import threading

import numpy as np

from ._abi import (KLT_MAX_LEVELS, KltAffineRec, KltBackendError, KltCommTimeout, KltOutOfMemory, KltFeat, KltKernelTime, KltParams,
                   KltTrackStats, load_library)
from .params import affine_params_from_tc, params_from_tc, taps_from_params

FEAT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("val", np.int32), ("aux", np.int32)])
assert FEAT_DTYPE.itemsize == C.sizeof(KltFeat)
AFFINE_DTYPE = np.dtype([("aff_x", np.float32), ("aff_y", np.float32), ("Axx", np.float32), ("Ayx", np.float32),
                         ("Axy", np.float32), ("Ayy", np.float32), ("valid", np.int32), ("pad", np.int32)])
assert AFFINE_DTYPE.itemsize == C.sizeof(KltAffineRec)

SELECTING_ALL = 1
REPLACING_SOME = 2
# the reference-shaped API's tracker reads and writes its pinned record arrays in place (klt_featbuf_map_host); KLT_MAP_RECORDS=0 in
# the environment goes back to one copy command each way
MAP_RECORDS = _os.environ.get("KLT_MAP_RECORDS", "1") != "0"
# feature buffers of the reference-shaped API (65534 / 65535 are the staging buffers of the synchronous klt_track / klt_select entry
# points, 65533 the API's selection list, 60000 .. 65524 KLTTrackSequence's table rows)
_FB_API_IN, _FB_API_OUT = 65526, 65527


def _dp(a):
    return (C.c_double * len(a))(*a)


class Context:
    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.klt_create(int(device), C.byref(h))
        if rc != 0:
            msg = self._lib.klt_last_error(None)
            raise KltBackendError("klt_create(device=%d) failed (%d): %s" % (device, rc, (msg or b"").decode()))
        self._h = h
        self.device = int(device)
        self._params_key = None
        # One context = one HIP stream set + one set of pinned record buffers: never two host threads inside it at a time
        # (include/klt_gpu.h, "Threads").  The reference-shaped API holds this lock for the length of a KLT* call; callers that
        # drive a Context directly from several threads take it themselves (`with ctx.lock:`) or give each thread its own.
        self.lock = threading.RLock()
        self._deferred = collections.deque()          # releases asked for by finalizers that could not get the lock at once

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.klt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            msg = self._lib.klt_last_error(self._h)
            kind = KltCommTimeout if rc == -5 else (KltOutOfMemory if rc == -4 else KltBackendError)      # KLT_ERR_TIMEOUT, KLT_ERR_NOMEM
            raise kind("libkltgpu error %d: %s" % (rc, (msg or b"").decode()))
        return rc

    def sync(self):
        self._check(self._lib.klt_sync(self._h))

    def stream_handle(self):
        return self._lib.klt_stream_handle(self._h)

    # ---------------------------------------------------------------- parameters
    def set_params(self, p):
        """p: KltParams.  Taps for the three sigmas are generated on the host (convolve.py:27-93)."""
        key = bytes(p)
        if key == self._params_key:
            return
        self._check(self._lib.klt_set_params(self._h, C.byref(p)))
        # taps depend on the three sigmas only: a change of mindist / max_residue / ... must not touch them (the library
        # invalidates resident pyramids when taps change -- sequentialMode keeps tc.pyramid_last across such changes)
        sig = (p.smooth_sigma, p.pyramid_sigma, p.grad_sigma)
        if sig != getattr(self, "_sigma_key", None):
            for which, (g, d) in enumerate(taps_from_params(p)):
                self._check(self._lib.klt_set_kernels(self._h, which, _dp(g), len(g), _dp(d), len(d)))
            self._sigma_key = sig
        self._params_key = key
        self.params = p

    def configured_for(self, tc):
        """True when configure(tc) would change nothing that voids resident pyramids"""
        return bytes(params_from_tc(tc)) == self._params_key

    def configure(self, tc):
        self.set_params(params_from_tc(tc))
        self.set_affine_params(affine_params_from_tc(tc))

    # ------------------------------------------------ affine consistency check
    def set_affine_params(self, ap):
        key = bytes(ap)
        if key != getattr(self, "_affine_key", None):
            self._check(self._lib.klt_set_affine_params(self._h, C.byref(ap)))
            self._affine_key = key

    def affine_alloc(self, state, n):
        self._check(self._lib.klt_affine_alloc(self._h, state, n))

    def affine_copy(self, dst, src, n, with_templates=False):
        """Device-side snapshot of the per-feature affine state (klt_affine_copy_async); asynchronous."""
        self._check(self._lib.klt_affine_copy_async(self._h, dst, src, n, int(bool(with_templates))))

    def affine_free(self, state):
        if getattr(self, "_h", None):
            self._check(self._lib.klt_affine_free(self._h, state))

    def slot_free(self, slot):
        if getattr(self, "_h", None):
            self._check(self._lib.klt_slot_free(self._h, slot))

    # ids handed to the reference-shaped API objects (tracking contexts: 3 slots; feature lists: affine state) are recycled
    # when those objects die (weakref finalizers in selectGoodFeatures.py / trackFeatures.py)
    def take_slots(self, count=3):
        free = self.__dict__.setdefault("_free_slot_bases", [])
        if free:
            return free.pop()
        base = getattr(self, "_next_slot", 0)
        self._next_slot = base + count
        return base

    def release_slots(self, base, count=3):
        """Finalizer of a tracking context (any thread, any moment -- the garbage collector runs it): never waits for the lock."""
        self._when_free(self._release_slots, base, count)

    def _release_slots(self, base, count):
        try:
            for k in range(count):
                self.slot_free(base + k)
        except KltBackendError:
            return
        self.__dict__.setdefault("_free_slot_bases", []).append(base)

    def _when_free(self, fn, *args):
        """Run fn(*args) under the context's lock if nobody holds it (or this thread does); otherwise leave it for the next holder
        (`settle_deferred`).  A finalizer that WAITED here could deadlock two threads whose collectors each free an object of the
        other's context."""
        if self.lock.acquire(blocking=False):
            try:
                fn(*args)
            finally:
                self.lock.release()
        else:
            self._deferred.append((fn, args))

    def settle_deferred(self):
        """(lock held) the releases finalizers left behind while another thread was inside the context"""
        while self._deferred:
            try:
                fn, args = self._deferred.popleft()
            except IndexError:
                return
            fn(*args)

    def take_affine_state(self):
        free = self.__dict__.setdefault("_free_affine_states", [])
        if free:
            return free.pop()
        sid = getattr(self, "_next_affine_state", 0)
        self._next_affine_state = sid + 1
        return sid

    def release_affine_state(self, sid):
        self._when_free(self._release_affine_state, sid)

    def _release_affine_state(self, sid):
        try:
            self.affine_free(sid)
        except KltBackendError:
            return
        self.__dict__.setdefault("_free_affine_states", []).append(sid)

    def affine_download(self, state, n):
        out = np.empty(n, AFFINE_DTYPE)
        self._check(self._lib.klt_affine_download(self._h, state, out.ctypes.data, n))
        return out

    def track_affine(self, slot1, slot2, fl, state):
        fl = np.ascontiguousarray(fl, FEAT_DTYPE).copy()
        k = C.c_int()
        self._check(self._lib.klt_track_affine(self._h, slot1, slot2, fl.ctypes.data, len(fl), state, C.byref(k)))
        return fl, k.value

    def track_affine_async(self, slot1, slot2, fb_in, fb_out, n, state):
        self._check(self._lib.klt_track_affine_async(self._h, slot1, slot2, fb_in, fb_out, n, state))

    # -------------------------------------------------------------------- frames
    def upload(self, slot, img):
        """img: 2-D uint8 or float32 array (the reference's `np.array(pil.convert("F"))`)."""
        a = np.asarray(img)
        if a.ndim != 2:
            raise ValueError("expected a 2-D image")
        if a.dtype == np.uint8:
            a = np.ascontiguousarray(a)
            self._check(self._lib.klt_upload_u8(self._h, slot, a.ctypes.data, a.shape[1], a.shape[0], a.shape[1]))
        else:
            a = np.ascontiguousarray(a, np.float32)
            self._check(self._lib.klt_upload_f32(self._h, slot, a.ctypes.data, a.shape[1], a.shape[0], a.shape[1]))

    def pinned_array(self, shape, dtype=np.uint8):
        """numpy array backed by pinned host memory (valid until the context is closed)."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.klt_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def upload_async(self, slot, img):
        """img: 2-D uint8 or float32 array from pinned_array(); enqueued on the copy stream, no host synchronisation."""
        if img.dtype not in (np.uint8, np.float32) or img.ndim != 2 or not img.flags["C_CONTIGUOUS"]:
            raise ValueError("upload_async takes a C-contiguous 2-D uint8 or float32 array in pinned memory")
        fn = self._lib.klt_upload_u8_async if img.dtype == np.uint8 else self._lib.klt_upload_f32_async
        self._check(fn(self._h, slot, img.ctypes.data, img.shape[1], img.shape[0], img.shape[1]))

    def device_alloc(self, nbytes):
        """Device memory owned by the context (an address as int); freed by device_free or with the context."""
        p = C.c_void_p()
        self._check(self._lib.klt_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def device_write(self, dev, arr):
        """Synchronous copy of a C-contiguous array to device address `dev`."""
        a = np.ascontiguousarray(arr)
        self._check(self._lib.klt_device_write(self._h, C.c_void_p(dev), a.ctypes.data, a.nbytes))

    def device_free(self, dev):
        self._check(self._lib.klt_device_free(self._h, C.c_void_p(dev)))

    def adopt_u8(self, slot, dev, ncols, nrows):
        """The slot's frame IS the u8 image at device address `dev` (contiguous rows): read in place, not copied (klt_slot_adopt_u8)."""
        self._check(self._lib.klt_slot_adopt_u8(self._h, slot, C.c_void_p(dev), ncols, nrows, ncols))

    def upload_wait(self):
        """Host waits until every upload_async issued so far has left its pinned source buffer."""
        self._check(self._lib.klt_upload_wait(self._h))

    def staging(self, shape, count=2, dtype=np.uint8):
        """`count` pinned staging buffers of `shape` (uint8, or float32 for colour / float frames), cached per context (pinned memory is
        never handed out twice)."""
        cache = self.__dict__.setdefault("_staging", {})
        key = (tuple(shape), count, np.dtype(dtype).char)
        if key not in cache:
            cache[key] = [self.pinned_array(shape, dtype) for _ in range(count)]
        return cache[key]

    def staging_forget(self, shape, count=2, dtype=np.uint8):
        """Drops a cached staging set without freeing it (somebody may still write to it); the next staging() allocates anew."""
        self.__dict__.setdefault("_staging", {}).pop((tuple(shape), count, np.dtype(dtype).char), None)

    def build_pyramids(self, slot, sync=True):
        fn = self._lib.klt_build_pyramids if sync else self._lib.klt_build_pyramids_async
        self._check(fn(self._h, slot))

    def build_pyramids_batch(self, slots, sync=False):
        arr = (C.c_int * len(slots))(*slots)
        self._check(self._lib.klt_build_pyramids_batch_async(self._h, arr, len(slots)))
        if sync:
            self.sync()

    def set_option(self, option, value):
        self._check(self._lib.klt_set_option(self._h, option, int(value)))

    def pyramids_valid(self, slot):
        """True while the slot's pyramids are built and match the current parameters."""
        return bool(self._check(self._lib.klt_slot_state(self._h, slot)) & 2)

    def frame_resident(self, slot):
        """True while the slot holds a frame (raw pixels on the device)."""
        return bool(self._check(self._lib.klt_slot_state(self._h, slot)) & 1)

    def device_memory(self):
        """(free, total) bytes of the context's device"""
        f, t = C.c_size_t(), C.c_size_t()
        self._check(self._lib.klt_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def slot_generation(self, slot):
        """Number of the build that filled the slot's pyramids (travels with swap_slots); 0 without valid pyramids."""
        g = C.c_uint64()
        self._check(self._lib.klt_slot_generation(self._h, slot, C.byref(g)))
        return g.value

    def swap_slots(self, a, b):
        self._check(self._lib.klt_swap_slots(self._h, a, b))

    def level_dims(self, slot, level):
        nc, nr = C.c_int(), C.c_int()
        self._check(self._lib.klt_level_dims(self._h, slot, level, C.byref(nc), C.byref(nr)))
        return nc.value, nr.value

    def download_level(self, slot, pyramid, level):
        nc, nr = self.level_dims(slot, level)
        out = np.empty((nr, nc), np.float32)
        self._check(self._lib.klt_download_f32(self._h, slot, pyramid, level, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ features
    def featbuf_upload(self, fb, fl):
        fl = np.ascontiguousarray(fl, FEAT_DTYPE)
        self._check(self._lib.klt_featbuf_upload(self._h, fb, fl.ctypes.data, len(fl)))

    def featbuf_download(self, fb, n):
        out = np.empty(n, FEAT_DTYPE)
        self._check(self._lib.klt_featbuf_download(self._h, fb, out.ctypes.data, n))
        return out

    def featbuf_download_into(self, fb, out):
        """Download out.size records of feature buffer `fb` straight into the C-contiguous record array `out`."""
        if out.dtype != FEAT_DTYPE or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("featbuf_download_into takes a C-contiguous array of klt_feat records")
        self._check(self._lib.klt_featbuf_download(self._h, fb, out.ctypes.data, out.size))

    def featbuf_download_async(self, fb, out):
        """Enqueue the download of len(out) records of feature buffer `fb` into the PINNED record array `out` (pinned_array(..., FEAT_DTYPE));
        the records are there after download_wait().  The host does not wait for queued work (klt_featbuf_download_async)."""
        self._check(self._lib.klt_featbuf_download_async(self._h, fb, out.ctypes.data, len(out)))

    def download_wait(self):
        self._check(self._lib.klt_download_wait(self._h))

    def featbuf_alloc(self, fb, n):
        self._check(self._lib.klt_featbuf_alloc(self._h, fb, n))

    def featbuf_view(self, fb_view, fb_parent, offset, n):
        self._check(self._lib.klt_featbuf_view(self._h, fb_view, fb_parent, offset, n))

    def featbuf_devptr(self, fb):
        return self._lib.klt_featbuf_devptr(self._h, fb)

    # ----------------------------------------------------------------- selection
    def select(self, slot, n, mode=SELECTING_ALL, fl=None, use_pyramid=False):
        """Returns (structured feature array, number placed)."""
        if fl is None:
            fl = np.zeros(n, FEAT_DTYPE)
            fl["x"] = -1
            fl["y"] = -1
            fl["val"] = -1
        fl = np.array(fl, FEAT_DTYPE)          # always a copy: the caller's array is never modified
        placed = C.c_int()
        self._check(self._lib.klt_select(self._h, slot, mode, int(bool(use_pyramid)), fl.ctypes.data, len(fl), C.byref(placed)))
        return fl, placed.value

    def min_distance_walk(self, keys, ncols, nrows, mindist, overwrite_all, fl):
        """klt_min_distance_walk: the greedy minimum-distance walk over the candidate keys in the given order; returns (a copy of
        `fl` with its free slots filled, number placed)."""
        keys = np.ascontiguousarray(keys, np.uint64)
        fl = np.array(fl, FEAT_DTYPE)
        placed = C.c_int()
        self._check(self._lib.klt_min_distance_walk(self._h, keys.ctypes.data if keys.size else None, int(keys.size), int(ncols),
                                                    int(nrows), int(mindist), int(bool(overwrite_all)), fl.ctypes.data, len(fl),
                                                    C.byref(placed)))
        return fl, placed.value

    def host_records(self, n):
        """(in, out): two pinned arrays of n klt_feat records -- the host side of the reference-shaped API's lists (no staging copy
        inside the runtime, and the copies can be asynchronous).  Valid until the next call that uses them.  Views of ONE pair of
        pinned arrays that only ever grows (to twice what was asked for): a script that alternates between lists of two lengths keeps
        one mapping of the tracker's feature buffers (`_map_records`; a pair per length was remapped -- with a device-wide wait each
        time -- whenever the length changed: ADVICE r5)."""
        pair = self.__dict__.get("_host_records")
        if pair is None or len(pair[0]) < n:
            if pair is not None:
                mapped = self.__dict__.get("_mapped_records")
                if mapped is not None:                                 # the tracker's feature buffers are these arrays: unmap first
                    for fb in mapped[2:]:
                        self._check(self._lib.klt_featbuf_map_host(self._h, fb, None, 0))
                    self._mapped_records = None
                for a in pair:
                    self._check(self._lib.klt_host_free(self._h, C.c_void_p(a.ctypes.data)))
                self._host_records = None
            cap = max(256, n if pair is None else 2 * n)
            pair = self._host_records = (self.pinned_array((cap,), FEAT_DTYPE), self.pinned_array((cap,), FEAT_DTYPE))
        return pair[0][:n], pair[1][:n]

    def select_records(self, slot, n, mode=SELECTING_ALL, use_pyramid=False, fb=None):
        """klt_select for the host API without its spare round trips: SELECTING_ALL needs no list on the way in (every slot is
        written), REPLACING_SOME sends host_records(n)[0] (filled by the caller) without waiting for the copy; one download brings
        the records back into host_records(n)[1], which is returned."""
        self.select_enqueue(slot, n, mode, use_pyramid, fb)
        return self.select_complete(n, fb)

    def select_enqueue(self, slot, n, mode=SELECTING_ALL, use_pyramid=False, fb=None):
        """First half of select_records: everything up to the host's look at the outcome is enqueued (klt_select_begin_async) on the
        list in host_records(n)[0] (filled by the caller for REPLACING_SOME); the caller does its own host work, then calls
        select_complete.  With MAP_RECORDS (the default) the feature buffer IS that pinned array -- the kernels update it in place, no
        copy either way; otherwise the list goes up first."""
        if fb is None:
            fb = _FB_API_IN if MAP_RECORDS else 65533
        if MAP_RECORDS and fb == _FB_API_IN:
            self._map_records(n, _FB_API_IN, _FB_API_OUT)
        elif mode == REPLACING_SOME:
            self._check(self._lib.klt_featbuf_upload_async(self._h, fb, self.host_records(n)[0].ctypes.data, n))
        self._check(self._lib.klt_select_begin_async(self._h, slot, mode, int(bool(use_pyramid)), fb, n))

    def select_complete(self, n, fb=None):
        """Second half: waits for the selection (klt_select_finish); the records are host_records(n)[0] itself when the buffer is
        mapped, else they come back into host_records(n)[1] with one download."""
        if fb is None:
            fb = _FB_API_IN if MAP_RECORDS else 65533
        self._check(self._lib.klt_select_finish(self._h))
        if MAP_RECORDS and fb == _FB_API_IN:
            # klt_select_finish has waited for the parallel passes' last launch -- but the sorted serial walk (no candidates at all, an
            # exclusion square too large for the passes' LDS tile, KLT_OPT_SELECT_PARALLEL_NMS = 0) completes inside
            # klt_select_begin_async WITHOUT a host wait: the download that used to follow was that wait.  The records are only in the
            # mapped array once the stream is idle (found by tests/fuzz/fuzz_seeds_r05.sh: 1 of 15 000 sequence trials; the regression
            # test fails three times out of three without this line).
            self._check(self._lib.klt_sync(self._h))
            return self.host_records(n)[0]
        rout = self.host_records(n)[1]
        self._check(self._lib.klt_featbuf_download(self._h, fb, rout.ctypes.data, n))
        return rout

    def track_records(self, slot1, slot2, n, state=None, fb_in=_FB_API_IN, fb_out=_FB_API_OUT):
        """klt_track / klt_track_affine on host_records(n): [0] (filled by the caller) goes up without waiting for the copy, the
        tracked records come back in [1], which is returned."""
        self.track_enqueue(slot1, slot2, n, state, True, fb_in, fb_out)
        return self.track_complete(n, fb_out)

    def _map_records(self, n, fb_in, fb_out):
        """feature buffers fb_in / fb_out ARE host_records(n) (klt_featbuf_map_host): the tracker reads and writes them in place"""
        self.host_records(n)                                       # (grows the pair if it has to -- and unmaps it first)
        rin, rout = self._host_records
        key = (rin.ctypes.data, rout.ctypes.data, fb_in, fb_out)
        if self.__dict__.get("_mapped_records") != key:
            self._check(self._lib.klt_featbuf_map_host(self._h, fb_in, rin.ctypes.data, len(rin)))
            self._check(self._lib.klt_featbuf_map_host(self._h, fb_out, rout.ctypes.data, len(rout)))
            self._mapped_records = key

    def track_enqueue(self, slot1, slot2, n, state=None, upload=True, fb_in=_FB_API_IN, fb_out=_FB_API_OUT):
        """The tracker is enqueued on the list in host_records(n)[0]; nothing is waited for.  With MAP_RECORDS (the default) the two
        feature buffers are those pinned arrays themselves and no copy is enqueued; otherwise the list goes up first (`upload`; not
        again when the tracker is only repeated on other pyramids)."""
        if MAP_RECORDS:
            self._map_records(n, fb_in, fb_out)
        elif upload:
            self._check(self._lib.klt_featbuf_upload_async(self._h, fb_in, self.host_records(n)[0].ctypes.data, n))
        if state is None:
            self._check(self._lib.klt_track_async(self._h, slot1, slot2, fb_in, fb_out, n))
        else:
            self._check(self._lib.klt_track_affine_async(self._h, slot1, slot2, fb_in, fb_out, n, state))

    def track_mark(self):
        """Marks the main stream behind the tracker just enqueued: track_complete(marked=True) then waits for THAT point, and work
        enqueued after the mark (the scores of the replacement that follows) keeps running while the host moves the columns."""
        if MAP_RECORDS:
            self._check(self._lib.klt_download_mark_async(self._h))
            return True
        return False

    def track_complete(self, n, fb_out=_FB_API_OUT, marked=False):
        """The records of the LAST tracker enqueued into fb_out, in host_records(n)[1] (a wait for the stream -- or for the mark --
        when the buffers are mapped, one synchronous download otherwise)."""
        rout = self.host_records(n)[1]
        if MAP_RECORDS and marked:
            self._check(self._lib.klt_download_wait(self._h))
        elif MAP_RECORDS:
            self._check(self._lib.klt_sync(self._h))
        else:
            self._check(self._lib.klt_featbuf_download(self._h, fb_out, rout.ctypes.data, n))
        return rout

    def select_async(self, slot, mode, use_pyramid, fb, n):
        self._check(self._lib.klt_select_async(self._h, slot, mode, int(bool(use_pyramid)), fb, n))

    def select_begin(self, slot, mode, use_pyramid, fb, n):
        """First half of select_async: everything enqueued up to the host's look at the outcome (klt_select_begin_async)."""
        self._check(self._lib.klt_select_begin_async(self._h, slot, mode, int(bool(use_pyramid)), fb, n))

    def select_finish(self):
        """Second half: waits, and completes the selection (klt_select_finish).  True when the list was rewritten after the first half's
        launches had run: work enqueued in between that read the list (a tracker launch of the next frame) must be enqueued again."""
        return self._check(self._lib.klt_select_finish(self._h)) > 0

    def select_prepare(self, slot):
        """Scores of the slot's level-0 images ahead of a REPLACING_SOME selection on it (klt_select_prepare_async); asynchronous."""
        self._check(self._lib.klt_select_prepare_async(self._h, slot))

    def select_intermediate(self, what):
        nc, nr = C.c_int(), C.c_int()
        self._check(self._lib.klt_select_dims(self._h, what, C.byref(nc), C.byref(nr)))
        out = np.empty((nr.value, nc.value), np.float32)
        self._check(self._lib.klt_download_select_f32(self._h, what, out.ctypes.data))
        return out

    def set_score_override(self, val):
        """Test hook: the next selection uses this [ny][nx] eigenvalue map instead of computing one."""
        val = np.ascontiguousarray(val, np.float32)
        self._check(self._lib.klt_set_score_override(self._h, val.ctypes.data, val.size))

    def sorted_candidates(self, n):
        val = np.empty(n, np.float32)
        x = np.empty(n, np.int32)
        y = np.empty(n, np.int32)
        nv = C.c_int()
        self._check(self._lib.klt_download_sorted_candidates(self._h, val.ctypes.data, x.ctypes.data, y.ctypes.data, n, C.byref(nv)))
        return val[:nv.value], x[:nv.value], y[:nv.value]

    # ------------------------------------------------------------------ tracking
    def track(self, slot1, slot2, fl):
        """In-place on a copy: returns (structured feature array, number still tracked)."""
        fl = np.ascontiguousarray(fl, FEAT_DTYPE).copy()
        k = C.c_int()
        self._check(self._lib.klt_track(self._h, slot1, slot2, fl.ctypes.data, len(fl), C.byref(k)))
        return fl, k.value

    def track_async(self, slot1, slot2, fb_in, fb_out, n):
        self._check(self._lib.klt_track_async(self._h, slot1, slot2, fb_in, fb_out, n))

    def track_batch_async(self, pairs, n):
        """pairs: [(slot1, slot2, fb_in, fb_out), ...] -- one tracker launch for all of them."""
        cols = [(C.c_int * len(pairs))(*[p[k] for p in pairs]) for k in range(4)]
        self._check(self._lib.klt_track_batch_async(self._h, cols[0], cols[1], cols[2], cols[3], len(pairs), n))

    def track_stats_reset(self):
        self._check(self._lib.klt_track_stats_reset(self._h))

    def track_stats(self):
        s = KltTrackStats()
        self._check(self._lib.klt_track_stats_read(self._h, C.byref(s)))
        return {"features": int(s.features),
                "level_visits": [int(v) for v in s.level_visits],
                "iterations": [int(v) for v in s.iterations]}

    # ------------------------------------------------------- multi-GPU (RCCL inside libkltgpu)
    def comm_init(self, nranks, rank, unique_id):
        """unique_id: the 128 bytes rank 0 got from klt_comm_unique_id (parallel.init_communicators exchanges them)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.klt_comm_init_rank(self._h, int(nranks), int(rank), buf))

    def comm_destroy(self):
        self._check(self._lib.klt_comm_destroy(self._h))

    def comm_info(self):
        n, r = C.c_int(), C.c_int()
        self._check(self._lib.klt_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def allgather_featbuf_async(self, fb_src, fb_dst, n):
        self._check(self._lib.klt_allgather_featbuf_async(self._h, fb_src, fb_dst, n))

    def gather_featbuf_async(self, fb_src, fb_dst, n, root=0):
        self._check(self._lib.klt_gather_featbuf_async(self._h, fb_src, fb_dst, n, root))

    def gatherv_featbuf_async(self, fb_src, fb_dst, counts, root=0):
        """Gather with a count per rank (shards of unequal size): rank r contributes counts[r] records (klt_gatherv_featbuf_async)."""
        arr = (C.c_int * len(counts))(*[int(v) for v in counts])
        self._check(self._lib.klt_gatherv_featbuf_async(self._h, fb_src, fb_dst, arr, root))

    def comm_set_timeout(self, ms):
        """Host-side waits on the communicator give up after `ms` milliseconds (KLT_ERR_TIMEOUT) instead of hanging."""
        self._check(self._lib.klt_comm_set_timeout(self._h, float(ms)))

    def sendrecv_featbuf(self, fb_send, to, fb_recv, frm, n):
        """The feature list as a baton (klt_sendrecv_featbuf_async): n records of fb_send to rank `to` and / or from rank `frm` into
        fb_recv (-1: no such side); the context's stream waits for the arrival."""
        self._check(self._lib.klt_sendrecv_featbuf_async(self._h, fb_send, to, fb_recv, frm, n))

    def comm_fence(self):
        """The context's stream waits (on the device) for the collectives issued so far."""
        self._check(self._lib.klt_comm_fence_async(self._h))

    def comm_fence_featbuf(self, fb):
        """... only for the last collective that read or wrote feature buffer `fb`."""
        self._check(self._lib.klt_comm_fence_featbuf_async(self._h, fb))

    def comm_wait(self):
        self._check(self._lib.klt_comm_wait(self._h))

    def comm_allreduce_max(self, values):
        """Element-wise max over all ranks of up to 16 floats; synchronous (doubles as a barrier)."""
        a = (C.c_double * len(values))(*values)
        self._check(self._lib.klt_comm_allreduce_max(self._h, a, len(values)))
        return list(a)

    # -------------------------------------------------- standalone convolutions
    def convolve_separate(self, img, horiz, vert):
        """_convolveSeparate (klt_convolve_separate_f32): any two tap lists"""
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(img)
        self._check(self._lib.klt_convolve_separate_f32(self._h, img.ctypes.data, img.shape[1], img.shape[0], _dp(horiz), len(horiz),
                                                        _dp(vert), len(vert), out.ctypes.data))
        return out

    def smooth(self, img, gauss):
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(img)
        self._check(self._lib.klt_smooth_f32(self._h, img.ctypes.data, img.shape[1], img.shape[0], _dp(gauss), len(gauss),
                                             out.ctypes.data))
        return out

    def pyramid(self, img, subsampling, nlevels, gauss):
        """[level 1, ..., level nlevels-1] of KLTPyramid.Compute (klt_pyramid_f32): every level smoothed and subsampled on the device."""
        img = np.ascontiguousarray(img, np.float32)
        dims, (nr, nc) = [], img.shape
        for _ in range(1, nlevels):
            nc, nr = nc // subsampling, nr // subsampling
            dims.append((nr, nc))
        out = np.empty(sum(r * c for r, c in dims), np.float32)
        self._check(self._lib.klt_pyramid_f32(self._h, img.ctypes.data, img.shape[1], img.shape[0], nlevels, subsampling, _dp(gauss), len(gauss),
                                              out.ctypes.data))
        levels, off = [], 0
        for r, c in dims:
            levels.append(out[off:off + r * c].reshape(r, c))
            off += r * c
        return levels

    def gradients(self, img, gauss, deriv):
        img = np.ascontiguousarray(img, np.float32)
        gx = np.empty_like(img)
        gy = np.empty_like(img)
        self._check(self._lib.klt_gradients_f32(self._h, img.ctypes.data, img.shape[1], img.shape[0], _dp(gauss), len(gauss),
                                                _dp(deriv), len(deriv), gx.ctypes.data, gy.ctypes.data))
        return gx, gy

    # -------------------------------------------------------------------- timing
    def timing_enable(self, on=True):
        """1 / True: an event pair around every launch; 2: the level-0 pyramid launch by its dispatch's own timestamps (klt_gpu.h)."""
        self._check(self._lib.klt_timing_enable(self._h, int(on)))

    def timing_read(self):
        buf = (KltKernelTime * 32)()
        n = self._check(self._lib.klt_timing_read(self._h, buf, 32))
        return [{"name": buf[i].name.decode(), "launches": int(buf[i].launches), "total_ms": float(buf[i].total_ms),
                 "bytes": float(buf[i].bytes)} for i in range(n)]


_tls = threading.local()


def default_context(device=None):
    """The calling THREAD's context for `device` (created on first use): the one the reference-shaped Python API works on.

    One context per host thread and device (include/klt_gpu.h, "Threads"; SURVEY section 8(b): no shared mutable globals): two
    threads that each track their own video run on their own HIP streams, parameters and pinned record buffers, side by side.
    A tracking context (`KLT_TrackingContext`) stays with the context of the thread that first used it (`context_of`), because its
    frames and pyramids live in that context's slots; every KLT* call holds that context's lock, so a tracking context handed to
    another thread -- or shared by two -- is served one call at a time."""
    import os
    if device is None:
        device = int(os.environ.get("KLT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = load_library().klt_device_count()
        if n > 0:
            device %= n
    table = _tls.__dict__.setdefault("contexts", {})
    ctx = table.get(device)
    if ctx is None or ctx._h is None:
        ctx = table[device] = Context(device)
    return ctx


def context_of(tc, device=None):
    """The context a tracking context is bound to: the calling thread's default context when `tc` is first used, the same one ever
    after (whichever thread calls)."""
    ctx = tc.__dict__.get("_klt_ctx")
    if ctx is None or ctx._h is None:
        ctx = tc.__dict__["_klt_ctx"] = default_context(device)
        tc.__dict__.pop("_klt_slots", None)                 # (slots of a context that was closed under it)
        tc.__dict__.pop("_klt_frames", None)
    return ctx


__all__ = ["Context", "default_context", "context_of", "FEAT_DTYPE", "SELECTING_ALL", "REPLACING_SOME", "KltBackendError", "KltCommTimeout", "KltOutOfMemory",
           "KLT_MAX_LEVELS", "KltParams"]
