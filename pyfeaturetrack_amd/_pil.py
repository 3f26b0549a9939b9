"""Pillow images without a conversion: the row table of an 8-bit, colour or float image, straight from Pillow's own storage.

The reference's callers hand PIL images to KLTSelectGoodFeatures / KLTTrackFeatures (selectGoodFeatures.py:190, trackFeatures.py:165,176:
`img.convert("F")`).  Making a numpy array of a mode-"L" image (`np.asarray(img)`: Pillow encodes the image into a bytes object, numpy
wraps it) costs 0.43 ms per 1080p image -- several times what a whole call costs on numpy frames.  Pillow keeps an image as an
`ImagingMemoryInstance` whose `image8` member is a table of row addresses; `Image.getim()` (Pillow >= 11) hands out a capsule named
"Pillow Imaging" with that struct's address.  `rows_of(img)` reads the table's address out of it; libkltgpu's klt_host_compare_rows /
klt_host_copy_rows / klt_host_sample_rows then compare the image with the copy a slot was filled from, stage it into pinned memory and
take the frame cache's 1024-pixel lattice from the rows directly.  Colour images ("RGB" / "RGBA" / "RGBX": 4-byte pixels behind `image32`) are
compared and kept as they are and turned into the float image `img.convert("F")` would give by klt_host_luma_rows (Pillow's own expression,
(float)(299 R + 587 G + 114 B) / 1000.0f); "F" images are float rows already.

The struct is Pillow's private layout, so nothing is assumed about it: the first use runs a SELF-CHECK on a small known image -- the
first 160 bytes of its struct are searched for the two consecutive ints that are its width and height and for a word that is the address
of a table of `height` row addresses whose rows reproduce the image's bytes; every address is looked up in /proc/self/maps before it is
read, so a wrong guess cannot fault.  The same is repeated for an image mapped onto a numpy array (`Image.fromarray`), whose rows live
in the array's memory.  If anything fails -- another Pillow, no getim(), no /proc -- `rows_of` returns None for every image and callers
take the array path as before.  KLT_NO_PIL_ROWS=1 in the environment switches the whole thing off."""
import ctypes as C
import os
import struct
import threading

_lock = threading.Lock()
_layout = None          # None: not probed yet; False: probe failed; else (offset of xsize, of ysize, of image8, of image32: the two row-table pointers)
_why = None             # why the probe failed (for `status()`)
_SCAN = 160
_CAPSULE = b"Pillow Imaging"

_get_pointer = C.pythonapi.PyCapsule_GetPointer
_get_pointer.restype = C.c_void_p
_get_pointer.argtypes = [C.py_object, C.c_char_p]
_is_valid = C.pythonapi.PyCapsule_IsValid
_is_valid.restype = C.c_int
_is_valid.argtypes = [C.py_object, C.c_char_p]


def _readable_ranges():
    out = []
    with open("/proc/self/maps") as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2 and parts[1].startswith("r"):
                lo, _, hi = parts[0].partition("-")
                out.append((int(lo, 16), int(hi, 16)))
    return out


def _struct_address(img):
    cap = img.getim()
    if type(cap).__name__ != "PyCapsule" or not _is_valid(cap, _CAPSULE):
        return None, None
    return _get_pointer(cap, _CAPSULE), cap


def _probe_one(img, data, w, h, ranges, bpp=1):
    """(offset of xsize, offset of ysize, offset of the row-table pointer) in img's struct, or a string saying what went wrong; bpp = bytes
    per pixel of the storage (1: image8, 4: image32)"""
    def readable(addr, n):
        return addr and any(lo <= addr and addr + n <= hi for lo, hi in ranges)

    p, cap = _struct_address(img)
    if not p:
        return "Image.getim() did not return a 'Pillow Imaging' capsule"
    if not readable(p, _SCAN):
        return "the image struct is not in readable memory"
    raw = C.string_at(p, _SCAN)
    ints = struct.unpack("<%di" % (_SCAN // 4), raw)
    size_at = [k * 4 for k in range(len(ints) - 1) if ints[k] == w and ints[k + 1] == h]
    if len(size_at) != 1:
        return "xsize / ysize found %d times in the struct" % len(size_at)
    for off in range(0, _SCAN, 8):
        table = struct.unpack_from("<Q", raw, off)[0]
        if not readable(table, 8 * h):
            continue
        rows = struct.unpack("<%dQ" % h, C.string_at(table, 8 * h))
        if all(readable(r, w * bpp) for r in rows) and b"".join(C.string_at(r, w * bpp) for r in rows) == data:
            return size_at[0], size_at[0] + 4, off
    return "no row table in the first %d bytes of the struct" % _SCAN


def _probe():
    global _why
    if os.environ.get("KLT_NO_PIL_ROWS") == "1":
        _why = "KLT_NO_PIL_ROWS=1"
        return False
    try:
        import numpy as np
        from PIL import Image
        if not hasattr(Image.Image, "getim"):
            _why = "this Pillow has no Image.getim()"
            return False
        ranges = _readable_ranges()
        w, h = 23, 5
        data = bytes((7 * i + 3) % 251 for i in range(w * h))
        own = Image.frombytes("L", (w, h), data)                       # storage allocated by Pillow
        arr = np.frombuffer(data, np.uint8).reshape(h, w).copy()
        mapped = Image.fromarray(arr)                                  # rows inside the array's memory (or a copy: either way the rows say)
        found = [_probe_one(im, data, w, h, ranges) for im in (own, mapped)]
        for f in found:
            if isinstance(f, str):
                _why = f
                return False
        if found[0] != found[1]:
            _why = "two images disagree about the layout: %r / %r" % tuple(found)
            return False
        # 4-byte pixels live behind the other row table (image32): a colour image and a float image must agree about it
        rgb = Image.frombytes("RGB", (w, h), bytes((11 * i + 5) % 253 for i in range(3 * w * h)))
        flt = Image.frombytes("F", (w, h), np.linspace(-3.0, 900.0, w * h).astype(np.float32).tobytes())
        found32 = [_probe_one(rgb, rgb.tobytes("raw", "RGBX"), w, h, ranges, 4), _probe_one(flt, flt.tobytes(), w, h, ranges, 4)]
        for f in found32:
            if isinstance(f, str):
                _why = "4-byte pixels: " + f
                return False
        if found32[0] != found32[1] or found32[0][:2] != found[0][:2] or found32[0][2] == found[0][2]:
            _why = "the 8-bit and the 32-bit images disagree about the layout: %r / %r / %r" % (found[0], found32[0], found32[1])
            return False
        # ... and the table is live storage, not a snapshot: a pixel written through Pillow shows in the rows
        own.putpixel((3, 2), 200)
        p, cap = _struct_address(own)
        table = C.c_void_p.from_address(p + found[0][2]).value
        row2 = C.c_void_p.from_address(table + 16).value
        if C.string_at(row2 + 3, 1) != b"\xc8":
            _why = "a pixel written with putpixel does not show in the rows"
            return False
        return found[0] + (found32[0][2],)
    except Exception as e:                                             # noqa: BLE001 -- whatever it is, the array path still works
        _why = "%s: %s" % (type(e).__name__, e)
        return False


def layout():
    """(offset of xsize, of ysize, of the 8-bit row table, of the 32-bit row table) once the self-check has passed, else False"""
    global _layout
    if _layout is None:
        with _lock:
            if _layout is None:
                _layout = _probe()
    return _layout


def status():
    """{"active": bool, "layout": ..., "why_not": ...} for logs and tests"""
    lay = layout()
    return {"active": bool(lay), "layout": lay or None, "why_not": None if lay else _why}


class Rows:
    """the row table of one image: `table` = address of nrows row addresses, each row `ncols` pixels of `kind` "u8" (1 byte), "rgbx"
    (4 bytes: R, G, B, pad / alpha) or "f32" (4 bytes); `keep` holds what keeps the storage alive for as long as this object lives
    (the image, its core object -- the owner of the rows -- and the capsule)"""
    __slots__ = ("table", "nrows", "ncols", "keep", "kind", "row_bytes")

    def __init__(self, table, nrows, ncols, keep, kind="u8"):
        self.table, self.nrows, self.ncols, self.keep, self.kind = table, nrows, ncols, keep, kind
        self.row_bytes = ncols * (1 if kind == "u8" else 4)


KINDS = {"L": "u8", "RGB": "rgbx", "RGBA": "rgbx", "RGBX": "rgbx", "F": "f32"}


def rows_of(img):
    """Rows of a mode-"L" / "RGB" / "RGBA" / "RGBX" / "F" Pillow image, or None (not such an image, self-check failed, the struct does
    not say what the image says)."""
    kind = KINDS.get(getattr(img, "mode", None))
    if kind is None:
        return None
    lay = _layout if _layout is not None else layout()
    if not lay:
        return None
    try:
        p, cap = _struct_address(img)                      # (getim() loads a lazily opened file first)
        if not p:
            return None
        w, h = img.size
        if C.c_int.from_address(p + lay[0]).value != w or C.c_int.from_address(p + lay[1]).value != h or w <= 0 or h <= 0:
            return None
        table = C.c_void_p.from_address(p + lay[2 if kind == "u8" else 3]).value
        # (the capsule does not own the storage: the core object does.  Holding IT keeps the rows valid even if the image is given another
        # core meanwhile -- paste / putpixel on an array-mapped image copy the storage first)
        return Rows(table, h, w, (img, getattr(img, "im", None), cap), kind) if table else None
    except Exception:                                       # noqa: BLE001
        return None
