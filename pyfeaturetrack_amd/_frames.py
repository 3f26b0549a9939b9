"""Which image each device slot of a tracking context holds, and the pinned staging ring frames travel through.

The reference converts both images and rebuilds both pyramids on every KLTTrackFeatures call (trackFeatures.py:146-196) -- example1's
ping-pong (example1.py:53-56) builds the same two pyramids 200 times.  Here a slot remembers the image it was filled from: the object
(weak reference), its size and mode, and a signature of 1024 pixels sampled on a lattice.  A call that names an image a slot already
holds -- with pyramids that still match the tracking context -- uploads and builds nothing for it.  An image modified IN PLACE between two
calls is recognised as new when one of the sampled pixels changed; a caller that edits other pixels of the same object and expects
them to be seen calls `KLTForgetFrames(tc)` (or sets KLT_NO_FRAME_CACHE=1 in the environment, which disables the cache).

Frames that do have to travel are copied into a ring of pinned staging buffers and go out with klt_upload_u8_async on the context's copy
stream: the host copy of the second frame of a pair runs while the first one's DMA is in flight, and the build waits for both on the
device.
"""
import os
import weakref

import numpy as np

_LATTICE = 32
_DISABLED = os.environ.get("KLT_NO_FRAME_CACHE") == "1"


def _signature(img):
    """bytes of a 32 x 32 lattice of the image's pixels (a few microseconds for either kind of image)"""
    if isinstance(img, np.ndarray):
        h, w = img.shape[:2]
        return img[::max(1, h // _LATTICE), ::max(1, w // _LATTICE)].tobytes()
    try:
        from PIL import Image
        return img.resize((_LATTICE, _LATTICE), Image.NEAREST).tobytes()
    except Exception:                                       # noqa: BLE001 -- an image type we cannot sample is never "the same"
        return None


class FrameKey:
    """identity + size + sampled content of one image"""
    __slots__ = ("ref", "size", "kind", "sig")

    def __init__(self, img):
        try:
            self.ref = weakref.ref(img)
        except TypeError:
            self.ref = None
        if isinstance(img, np.ndarray):
            self.size, self.kind = (img.shape[1], img.shape[0]), img.dtype.char
        else:
            self.size, self.kind = tuple(img.size), getattr(img, "mode", "?")
        self.sig = _signature(img)

    def same_image(self, img, other):
        """`other` = the FrameKey just made of `img`"""
        return (self.ref is not None and self.ref() is img and self.sig is not None and self.size == other.size
                and self.kind == other.kind and self.sig == other.sig)


class FrameCache:
    """Per tracking context: what its device slots hold.  All methods run on the calling thread."""

    def __init__(self):
        self.held = {}                   # slot -> FrameKey

    def forget(self, slot=None):
        if slot is None:
            self.held.clear()
        else:
            self.held.pop(slot, None)

    def swap(self, a, b):
        ka, kb = self.held.pop(a, None), self.held.pop(b, None)
        if ka is not None:
            self.held[b] = ka
        if kb is not None:
            self.held[a] = kb

    def find(self, img, key, slots, ctx):
        """slot among `slots` that holds `img` as a frame (raw pixels resident), or None"""
        if _DISABLED:
            return None
        for s in slots:
            k = self.held.get(s)
            if k is not None and k.same_image(img, key) and ctx.frame_resident(s):
                return s
        return None

    def note(self, slot, key):
        if not _DISABLED:
            self.held[slot] = key


def cache_of(tc):
    c = getattr(tc, "_klt_frames", None)
    if c is None:
        c = tc._klt_frames = FrameCache()
    return c


def KLTForgetFrames(tc):
    """The next call on `tc` uploads and rebuilds every image it is given, whatever the device slots hold."""
    cache_of(tc).forget()


class Stager:
    """Pinned staging ring of a context for frames of one size (u8): `put(slot, array)` copies into the next buffer and enqueues the
    asynchronous upload.  A buffer is reused only after the copies issued from it have finished (klt_upload_wait: the copy stream
    only; kernels keep running)."""

    def __init__(self, ctx, shape, count=5):
        self.ctx, self.shape = ctx, tuple(shape)
        self.bufs = ctx.staging(self.shape, count=count)
        self.next = 0
        self.in_flight = 0

    def put(self, slot, arr):
        if self.in_flight >= len(self.bufs):
            self.ctx.upload_wait()
            self.in_flight = 0
        buf = self.bufs[self.next]
        self.next = (self.next + 1) % len(self.bufs)
        np.copyto(buf, arr)
        self.ctx.upload_async(slot, buf)
        self.in_flight += 1

    def settle(self):
        """every staged frame has left its buffer (call once the results of the work that consumed them are back: free by then)"""
        if self.in_flight:
            self.ctx.upload_wait()
            self.in_flight = 0


def stager_of(ctx, shape):
    table = ctx.__dict__.setdefault("_stagers", {})
    st = table.get(tuple(shape))
    if st is None:
        st = table[tuple(shape)] = Stager(ctx, shape)
    return st


def settle_frames(ctx, shape):
    """the staged frames of this size have left their pinned buffers (no-op when none were staged)"""
    st = ctx.__dict__.get("_stagers", {}).get(tuple(shape))
    if st is not None:
        st.settle()


def send_frame(ctx, slot, arr):
    """frame -> slot: u8 frames through the pinned ring (asynchronous), anything else with the synchronous upload"""
    if arr.dtype == np.uint8 and arr.ndim == 2:
        stager_of(ctx, arr.shape).put(slot, arr)
    else:
        ctx.upload(slot, arr)
