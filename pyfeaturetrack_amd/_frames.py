"""Which image each device slot of a tracking context holds, and the pinned host copies frames travel through.

The reference converts both images and rebuilds both pyramids on every KLTTrackFeatures call (trackFeatures.py:146-196) -- example1's
ping-pong (example1.py:53-56) builds the same two pyramids 200 times.  Here a slot remembers the frame it was filled from by keeping
the host copy the upload was made from (pinned memory for 8-bit frames -- the DMA source anyway).  A call whose image has EXACTLY the
pixels a slot already holds -- every byte is compared, after a 1024-pixel lattice has served as a fast reject -- uploads and builds
nothing for it; any other image, including the same object edited in place by one pixel, is sent and rebuilt, as the reference would.
The results are therefore those of the reference for any call sequence; what the cache saves is the upload and the pyramid build, what
it costs is one pass over the frame on the host (2 MB at 1080p: about as long as the copy into pinned memory it replaces).

Opt-in shortcut, a documented deviation: `tc.trustFrameIdentity = True` (or KLT_TRUST_FRAME_IDENTITY=1 in the
environment) trusts object identity + size + the lattice and skips the full comparison -- for callers that never edit an image in
place, or call KLTForgetFrames(tc) when they do.  KLT_NO_FRAME_CACHE=1 disables the cache altogether.
"""
import ctypes
import os
import weakref

import numpy as np

from ._pil import rows_of

_LATTICE = 32
_DISABLED = os.environ.get("KLT_NO_FRAME_CACHE") == "1"
_TRUST_ENV = os.environ.get("KLT_TRUST_FRAME_IDENTITY") == "1"

_host = None


def _host_lib():
    """libkltgpu's host-side helpers (klt_host_compare / klt_host_copy: a frame spread over a few parked worker threads)"""
    global _host
    if _host is None:
        from ._abi import load_library
        _host = load_library()
    return _host


def _lattice(arr):
    h, w = arr.shape[:2]
    return arr[::max(1, h // _LATTICE), ::max(1, w // _LATTICE)].tobytes()


def same_pixels(a, b):
    """every byte of two arrays of equal shape and dtype (klt_host_compare when both are contiguous)"""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.flags["C_CONTIGUOUS"] and b.flags["C_CONTIGUOUS"]:
        return _host_lib().klt_host_compare(a.ctypes.data, b.ctypes.data, a.nbytes) == 0
    return bool(np.array_equal(a, b))


def copy_pixels(dst, src):
    """src -> dst (equal shape and dtype; klt_host_copy when both are contiguous)"""
    if src.flags["C_CONTIGUOUS"] and dst.flags["C_CONTIGUOUS"] and src.dtype == dst.dtype and src.shape == dst.shape:
        _host_lib().klt_host_copy(dst.ctypes.data, src.ctypes.data, src.nbytes)
    else:
        np.copyto(dst, src)


def _to_array(img):
    """uint8 or float32 2-D array holding exactly `np.array(img.convert("F"))` (selectGoodFeatures.image_to_array)"""
    from .selectGoodFeatures import image_to_array
    return image_to_array(img)


class FrameKey:
    """One image as a call names it: the object, its size and mode; the pixel array and its lattice are made when first asked for
    (a Pillow image is converted once per call, and not at all in the trusting mode when identity or size already differ).
    A Pillow image -- what the reference's callers pass (trackFeatures.py:165,176) -- is not converted by Pillow at all: `rows` is the
    row table of its own storage (_pil.py), which the lattice, the comparison and the staging copy read in place: 8-bit images ("L") as
    they are, colour images ("RGB" / "RGBA" / "RGBX") compared and kept as 4-byte pixels and turned into the float frame
    `img.convert("F")` would be by klt_host_luma_rows, float images ("F") as float rows."""
    __slots__ = ("img", "size", "kind", "rows", "_arr", "_sig")

    def __init__(self, img):
        self.img = img
        self.rows = None
        if isinstance(img, np.ndarray):
            self.size, self.kind = (img.shape[1], img.shape[0]), img.dtype.char
        else:
            self.size, self.kind = tuple(img.size), getattr(img, "mode", "?")
            self.rows = rows_of(img)
        self._arr = self._sig = None

    def array(self):
        """uint8 or float32 2-D array holding exactly `np.array(img.convert("F"))`"""
        if self._arr is None:
            r = self.rows
            if r is None or r.kind == "u8":
                self._arr = _to_array(self.img)
            else:                                           # colour / float rows: Pillow's conversion without Pillow's intermediate image
                self._arr = np.empty((r.nrows, r.ncols), np.float32)
                self.float_into(self._arr)
        return self._arr

    def sig(self):
        if self._sig is None:
            r = self.rows
            if r is not None:
                # (4-byte pixels: the first byte of every sampled pixel -- R, or the float's low byte: a fast reject, nothing more)
                bpp = r.row_bytes // max(1, r.ncols)
                ys, xs = max(1, r.nrows // _LATTICE), max(1, r.ncols // _LATTICE)
                out = (ctypes.c_ubyte * (((r.nrows + ys - 1) // ys) * ((r.ncols + xs - 1) // xs)))()
                n = _host_lib().klt_host_sample_rows(r.table, r.nrows, r.row_bytes, ys, xs * bpp, out, len(out))
                self._sig = bytes(out) if n == len(out) else _lattice(self.array())
            else:
                self._sig = _lattice(self.array())
        return self._sig

    def same_as(self, kept):
        """every byte of the image against `kept`, the host copy a slot was filled from (8-bit pixels, 4-byte colour pixels or floats,
        as the image stores them)"""
        r = self.rows
        if r is not None and kept.flags["C_CONTIGUOUS"] and kept.shape[0] == r.nrows and kept.nbytes == r.nrows * r.row_bytes \
                and kept.dtype == (np.float32 if r.kind == "f32" else np.uint8):
            return _host_lib().klt_host_compare_rows(r.table, r.nrows, r.row_bytes, kept.ctypes.data) == 0
        return same_pixels(self.array(), kept)

    def stage_kind(self):
        """("u8" | "rgbx" | "f32", (nrows, ncols)) when the image can be staged into pinned memory and sent asynchronously, else None"""
        if self.rows is not None:
            return self.rows.kind, (self.rows.nrows, self.rows.ncols)
        arr = self.array()
        return ("u8", arr.shape) if arr.dtype == np.uint8 and arr.ndim == 2 else None

    def stage_u8(self):
        """(nrows, ncols) when the image can be staged as an 8-bit frame (`copy_into`), else None"""
        k = self.stage_kind()
        return k[1] if k is not None and k[0] == "u8" else None

    def copy_into(self, buf):
        """the image's pixels AS IT STORES THEM -> the contiguous buffer `buf` (pinned memory: the DMA's source for 8-bit and float
        images, the copy later calls are compared with for colour images)"""
        r = self.rows
        if r is not None and buf.flags["C_CONTIGUOUS"] and buf.nbytes == r.nrows * r.row_bytes:
            _host_lib().klt_host_copy_rows(buf.ctypes.data, r.table, r.nrows, r.row_bytes)
        else:
            copy_pixels(buf, self.array())

    def float_into(self, buf):
        """`img.convert("F")` of a colour or float Pillow image -> the contiguous float32 buffer `buf` [nrows][ncols]"""
        r = self.rows
        assert r is not None and r.kind != "u8" and buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"] and buf.shape == (r.nrows, r.ncols)
        if r.kind == "rgbx":
            _host_lib().klt_host_luma_rows(buf.ctypes.data, r.table, r.nrows, r.ncols)
        else:
            _host_lib().klt_host_copy_rows(buf.ctypes.data, r.table, r.nrows, r.row_bytes)


def pixels_of(img):
    """the image as a uint8 / float32 array of its own (`image_to_array`; a Pillow image is copied / converted out of its row table)"""
    key = FrameKey(img)
    if key.rows is None or key.rows.kind != "u8":
        return key.array()
    out = np.empty((key.rows.nrows, key.rows.ncols), np.uint8)
    key.copy_into(out)
    return out


class _Held:
    """what a slot holds: the host copy it was uploaded from (`kept`; pinned for 8-bit frames), the lattice of that copy, a weak
    reference to the image object, and the number of the last asynchronous upload made from `kept`"""
    __slots__ = ("ref", "size", "kind", "sig", "kept", "upload_no", "pool")

    def __init__(self, key, kept, upload_no, pool):
        try:
            self.ref = weakref.ref(key.img)
        except TypeError:
            self.ref = None
        self.size, self.kind, self.sig = key.size, key.kind, key.sig()
        self.kept, self.upload_no, self.pool = kept, upload_no, pool

    def release(self):
        """the pinned copy goes back to its context's pool (still carrying the number of its last upload)"""
        if self.pool is not None and self.kept is not None:
            self.pool.append((self.kept, self.upload_no))
        self.kept = self.pool = None

    def __del__(self):
        try:
            self.release()
        except Exception:                                   # noqa: BLE001 -- interpreter shutdown
            pass


class FrameCache:
    """Per tracking context: what its device slots hold.  All methods run on the calling thread."""

    def __init__(self, tc=None):
        self.held = {}                   # slot -> _Held
        self.handles = weakref.WeakSet()
        self._tc = weakref.ref(tc) if tc is not None else None

    def watch(self, handles):
        """pyramid handles (trackFeatures._ResidentPyramids) whose planes live in this context's slots"""
        for h in handles:
            self.handles.add(h)

    def keep_handles(self, ctx, slot):
        """`slot` is about to be overwritten: handles somebody still holds on the pyramids in it download their planes first (the
        reference's pyramid objects stay valid for as long as they are referenced)"""
        if self.handles:
            gen = ctx.slot_generation(slot)
            for h in list(self.handles):
                if gen and h._gen == gen:
                    h._materialise()
                    self.handles.discard(h)

    def keep_all_handles(self):
        """every live handle downloads its planes now (the parameters are about to change, which voids every pyramid of the context,
        or a sequence call is about to take the slots over)"""
        for h in list(self.handles):
            try:
                h._materialise()
            except RuntimeError:                            # its planes are gone already; the handle says so when asked
                pass
            self.handles.discard(h)

    def trusting(self):
        tc = self._tc() if self._tc is not None else None
        return _TRUST_ENV or bool(getattr(tc, "trustFrameIdentity", False))

    def forget(self, slot=None):
        for s in (list(self.held) if slot is None else [slot]):
            h = self.held.pop(s, None)
            if h is not None:
                h.release()

    def swap(self, a, b):
        ha, hb = self.held.pop(a, None), self.held.pop(b, None)
        if ha is not None:
            self.held[b] = ha
        if hb is not None:
            self.held[a] = hb

    def filled_from(self, key, slot):
        """the slot was last filled from this very image object (says nothing about its pixels now)"""
        h = self.held.get(slot)
        return h is not None and h.ref is not None and h.ref() is key.img

    def plausible(self, key, slot, ctx):
        """the fast rejects only: `slot` holds a frame of the image's size and mode with the same 1024-pixel lattice (and, in the
        trusting mode, filled from the very same object).  True is a candidate, not a match: `verify` decides (the trusting mode
        stops here, that is what it trusts)."""
        if _DISABLED:
            return False
        h = self.held.get(slot)
        if h is None or h.size != key.size or h.kind != key.kind or not ctx.frame_resident(slot):
            return False
        if self.trusting():
            return h.ref is not None and h.ref() is key.img and h.sig == key.sig()
        return h.sig == key.sig()

    def verify(self, key, slot):
        """every byte of the image against the host copy the slot was filled from (the trusting mode does not look)"""
        if self.trusting():
            return True
        h = self.held.get(slot)
        return h is not None and key.same_as(h.kept)

    def find(self, key, slots, ctx):
        """slot among `slots` whose resident frame has exactly the pixels of the image `key` names, or None"""
        for s in slots:
            if self.plausible(key, s, ctx) and self.verify(key, s):
                return s
        return None

    def send(self, ctx, slot, key):
        """frame -> slot, remembered: 8-bit frames are copied into the slot's pinned buffer and leave with klt_upload_u8_async on
        the context's copy stream (the host copy of the second frame of a pair runs while the first one's DMA is in flight, the
        build waits for both on the device); anything else goes with the synchronous upload and an ordinary copy is kept."""
        staged = key.stage_kind()
        self.keep_handles(ctx, slot)
        old = self.held.pop(slot, None)
        if staged is not None and hasattr(ctx, "upload_async"):
            kind, shape = staged
            # the copy later calls are compared with: the pixels as the image stores them (for 8-bit and float images also the DMA's source)
            kshape, kdtype = ((shape[0], 4 * shape[1]), np.uint8) if kind == "rgbx" else (shape, np.float32 if kind == "f32" else np.uint8)
            pool = _pool_of(ctx, kshape, kdtype)
            if old is not None and old.pool is pool and old.kept is not None:
                buf, no = old.kept, old.upload_no
                old.kept = old.pool = None
            elif pool:
                buf, no = pool.pop()
            else:
                buf, no = ctx.pinned_array(kshape, kdtype), 0
            if old is not None:
                old.release()
            _uploads_finished(ctx, no)                      # the DMA that last read this buffer has finished
            if kind == "rgbx":
                # a colour image: what travels is the float frame `img.convert("F")` would be, made in a pinned buffer of its own that goes
                # back to its pool at once (with the number of the upload that reads it: the next user waits for that copy); the copy of
                # the pixels as stored -- what later calls are compared with -- is made while that DMA runs
                fpool = _pool_of(ctx, shape, np.float32)
                fbuf, fno = fpool.pop() if fpool else (ctx.pinned_array(shape, np.float32), 0)
                _uploads_finished(ctx, fno)
                key.float_into(fbuf)
                ctx.upload_async(slot, fbuf)
                ctx.__dict__["_uploads_issued"] = issued = ctx.__dict__.get("_uploads_issued", 0) + 1
                fpool.append((fbuf, issued))
                key.copy_into(buf)
                no = 0                                      # (`buf` itself is never a DMA's source)
            else:
                key.copy_into(buf)
                ctx.upload_async(slot, buf)
                ctx.__dict__["_uploads_issued"] = no = ctx.__dict__.get("_uploads_issued", 0) + 1
            kept = buf
        else:
            if old is not None:
                old.release()
            arr = key.array()
            if hasattr(ctx, "upload"):
                ctx.upload(slot, arr)
            kept, no, pool = (arr if arr is not key.img else arr.copy()), 0, None
        if not _DISABLED:
            self.held[slot] = _Held(key, kept, no, pool)
        elif pool is not None:
            pool.append((kept, no))


def _pool_of(ctx, shape, dtype=np.uint8):
    """pinned frame buffers of this shape and type nobody holds at the moment: [(buffer, number of its last upload)]"""
    return ctx.__dict__.setdefault("_frame_pool", {}).setdefault((tuple(shape), np.dtype(dtype).char), [])


def _uploads_finished(ctx, upload_no):
    """host waits until asynchronous upload number `upload_no` of this context has left its source buffer (klt_upload_wait: the copy
    stream only; kernels keep running) -- no call at all when an earlier wait already covered it"""
    if upload_no > ctx.__dict__.get("_uploads_done", 0):
        issued = ctx.__dict__.get("_uploads_issued", 0)
        ctx.upload_wait()
        ctx.__dict__["_uploads_done"] = issued


def cache_of(tc):
    c = getattr(tc, "_klt_frames", None)
    if c is None:
        c = tc._klt_frames = FrameCache(tc)
    return c


def KLTForgetFrames(tc):
    """The next call on `tc` uploads and rebuilds every image it is given, whatever the device slots hold."""
    cache_of(tc).forget()


def settle_frames(ctx):
    """the results of the work that consumed the staged frames are back: every upload issued so far has left its buffer (the tracker
    / selection waited for the builds, which waited for the copies)"""
    ctx.__dict__["_uploads_done"] = ctx.__dict__.get("_uploads_issued", 0)
