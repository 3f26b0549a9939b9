"""Deterministic synthetic frames for the benchmark configs and the parity tests.

The reference ships only img0.pgm / img1.pgm (320x240).  The larger
BASELINE.json configs (1080p, 720p batches, 2160p sequences) use a seeded
texture with a known per-frame translation so that tracking has a known answer
(SURVEY.md section 8(d)).

The generator is numpy-only and bit-reproducible across machines:
  * `numpy.random.default_rng(seed).random` (PCG64) is platform independent;
  * the blur is a *circular* separable Gaussian built from `np.roll` and
    elementwise multiply/add (no reductions, no FMA fusion);
  * the sub-pixel shift is a bilinear blend of four rolled copies.
Because the texture is periodic, frame k of any length sequence is defined:
    frame_k(x, y) = base((x - k*sx) mod W, (y - k*sy) mod H)
so every feature moves by exactly (+sx, +sy) pixels per frame.
"""
import math

import numpy as np

DEFAULT_SHIFT = (3.3, -2.1)


def _circular_blur(a, sigma):
    r = int(math.ceil(3.0 * sigma))
    taps = [math.exp(-(i * i) / (2.0 * sigma * sigma)) for i in range(-r, r + 1)]
    s = sum(taps)
    taps = [t / s for t in taps]
    for axis in (1, 0):
        acc = np.zeros_like(a)
        for i, t in zip(range(-r, r + 1), taps):
            acc = acc + np.roll(a, i, axis=axis) * t
        a = acc
    return a


def synth_base(width, height, seed, sigma=2.0):
    """Periodic float64 texture in [0, 255] of shape (height, width)."""
    rng = np.random.default_rng(seed)
    a = rng.random((height, width))
    a = _circular_blur(a, sigma)
    lo = a.min()
    hi = a.max()
    return (a - lo) * (255.0 / (hi - lo))


def shift_frame(base, dx, dy):
    """uint8 frame whose content is `base` moved by (+dx, +dy) pixels (periodic)."""
    # frame(x, y) = base(x - dx, y - dy); split the shift into integer + fraction
    fx = math.floor(-dx)
    ax = (-dx) - fx
    fy = math.floor(-dy)
    ay = (-dy) - fy
    # b00(x, y) = base(x + fx, y + fy)
    b00 = np.roll(base, (-fy, -fx), axis=(0, 1))
    b01 = np.roll(b00, -1, axis=1)
    b10 = np.roll(b00, -1, axis=0)
    b11 = np.roll(b10, -1, axis=1)
    out = (b00 * ((1.0 - ax) * (1.0 - ay)) + b01 * (ax * (1.0 - ay))
           + b10 * ((1.0 - ax) * ay) + b11 * (ax * ay))
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def synth_frame(width, height, seed, k, shift=DEFAULT_SHIFT, base=None):
    """Frame k of the sequence for (width, height, seed); uint8 [height, width]."""
    if base is None:
        base = synth_base(width, height, seed)
    return shift_frame(base, k * shift[0], k * shift[1])


def sequence_phases(width, height, seed, shift=DEFAULT_SHIFT, base=None, workers=1):
    """frames 0..9 of the sequence: the ten sub-pixel phases periodic_sequence rolls (computed side by side with `workers` threads)"""
    if base is None:
        base = synth_base(width, height, seed)
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(workers, 10)) as ex:
            return list(ex.map(lambda k: synth_frame(width, height, seed, k, shift=shift, base=base), range(10)))
    return [synth_frame(width, height, seed, k, shift=shift, base=base) for k in range(10)]


def periodic_sequence(width, height, seed, count, shift=DEFAULT_SHIFT, base=None, phases=None, start=0):
    """Generator of `count` frames of a LONG sequence (BASELINE cfg-5: 512 frames of 3840x2160) at a few milliseconds per frame.
    With a shift whose tenfold is a whole number of pixels (the default 3.3, -2.1) the sub-pixel phase of frame k depends on k % 10
    only: frames 0..9 are synth_frame's, frame k >= 10 is frame k % 10 rolled by (k // 10) * (10 sx, 10 sy) whole pixels -- the same
    periodic texture moved by exactly k * (sx, sy).  (synth_frame(k) itself evaluates k * sx in floating point, whose fractional part can
    differ from that of (k % 10) * sx in the last bit: a frame of this generator is defined by the generator, not by synth_frame.)
    `phases` = sequence_phases(...) computed once for several passes over the sequence; `start` = first frame to yield."""
    sx10, sy10 = round(10 * shift[0]), round(10 * shift[1])
    if abs(10 * shift[0] - sx10) > 1e-9 or abs(10 * shift[1] - sy10) > 1e-9:
        raise ValueError("periodic_sequence needs a shift whose tenfold is a whole number of pixels")
    if phases is None and base is None:
        base = synth_base(width, height, seed)
    made = list(phases) if phases is not None else []
    for k in range(start, count):
        while len(made) <= k % 10:                          # `phases` (sequence_phases) given: nothing to compute
            made.append(synth_frame(width, height, seed, len(made), shift=shift, base=base))
        yield made[k] if k < 10 else np.roll(made[k % 10], ((k // 10) * sy10, (k // 10) * sx10), axis=(0, 1))


def synth_pair(width, height, seed, shift=DEFAULT_SHIFT):
    base = synth_base(width, height, seed)
    return shift_frame(base, 0.0, 0.0), shift_frame(base, shift[0], shift[1])


def warp_frame(base, A, t=(0.0, 0.0)):
    """uint8 frame whose content is `base` mapped by p -> c + A (p - c) + t about the frame centre c
    (periodic texture, bilinear resampling).  Known answer for the affine consistency check: a feature at p in
    `base` is found at c + A (p - c) + t, and its neighbourhood is deformed by the 2x2 matrix A."""
    h, w = base.shape
    A = np.asarray(A, np.float64)
    Ai = np.linalg.inv(A)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    dx, dy = xs - cx - t[0], ys - cy - t[1]
    sx = Ai[0, 0] * dx + Ai[0, 1] * dy + cx          # source position in `base`
    sy = Ai[1, 0] * dx + Ai[1, 1] * dy + cy
    x0 = np.floor(sx)
    y0 = np.floor(sy)
    ax, ay = sx - x0, sy - y0
    x0 = x0.astype(np.int64) % w
    y0 = y0.astype(np.int64) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    out = (base[y0, x0] * ((1 - ax) * (1 - ay)) + base[y0, x1] * (ax * (1 - ay))
           + base[y1, x0] * ((1 - ax) * ay) + base[y1, x1] * (ax * ay))
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
