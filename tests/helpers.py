"""Shared helpers for the parity tests."""
import hashlib

import numpy as np

from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc


def sha_bytes(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def make_tc(levels=None, ss=None, window=None, **attrs):
    tc = KLT_TrackingContext()
    if window is not None:
        tc.window_width = tc.window_height = window
    if levels is not None:
        tc.nPyramidLevels = levels
        tc.subsampling = ss
    tc.KLTUpdateTCBorder()
    for k, v in attrs.items():
        setattr(tc, k, v)
    return tc


def synth251_frames():
    base = synth.synth_base(251, 187, 11)
    return [synth.synth_frame(251, 187, 11, k, shift=(1.3, -0.8), base=base) for k in range(3)]


def feats_equal(fl, gx, gy, gv):
    """structured feature array vs golden (x, y as float64, val)"""
    return (np.array_equal(fl["val"].astype(np.int64), gv)
            and np.array_equal(fl["x"].astype(np.float64), gx)
            and np.array_equal(fl["y"].astype(np.float64), gy))


__all__ = ["sha_bytes", "make_tc", "synth251_frames", "feats_equal", "params_from_tc"]
