"""Frames on their way in and records on their way out: slots that adopt frames already in device memory, uploads
that stay ordered, frames too large for a plane, feature buffers mapped into pinned host memory, records read back without draining
the pipeline.  (Folded by component from the round-3 / 4 / 5 files in round 6: the tests are unchanged.)"""
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


def test_a_slot_adopts_a_frame_that_is_already_on_the_device(img0, img1, cfg1):
    """klt_slot_adopt_u8 (zero-copy ingest): a clip kept in device memory (klt_device_alloc / klt_device_write) is read in
    place -- the slot that adopts a frame gives the reference's pyramid planes, selection and tracking, exactly as the slot that was
    uploaded to; an upload into the slot ends the adoption, freeing the clip leaves no slot pointing into it, and the argument checks hold."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(max_residue=10.0))
        n = img0.size
        clip = c.device_alloc(2 * n)
        c.device_write(clip, img0)
        c.device_write(clip + n, img1)
        c.adopt_u8(0, clip, 320, 240)
        c.adopt_u8(1, clip + n, 320, 240)
        assert c.frame_resident(0) and not c.pyramids_valid(0)
        c.build_pyramids_batch([0, 1], sync=True)
        for slot, name in ((0, "p0"), (1, "p1")):
            for l in range(2):
                for pi, w in enumerate(("img", "gx", "gy")):
                    assert np.array_equal(c.download_level(slot, pi, l), cfg1["%s_%s_%d" % (name, w, l)]), (name, w, l)
        fl, placed = c.select(0, 100)                              # from the raw (adopted) frame, not the pyramid
        assert placed == 100 and np.array_equal(fl["x"].astype(np.float64), cfg1["sel100_x"]) and np.array_equal(fl["val"].astype(np.int64), cfg1["sel100_val"])
        out, _ = c.track(0, 1, fl)
        assert np.array_equal(out["val"].astype(np.int64), cfg1["trk100_r10_val"])
        ok = out["val"] >= 0
        assert np.array_equal(out["x"][ok].astype(np.float64), cfg1["trk100_r10_x"][ok])
        c.upload(0, img1)                                          # an upload ends the adoption: the clip's first frame is untouched
        c.adopt_u8(2, clip, 320, 240)
        c.build_pyramids_batch([0, 2], sync=True)
        assert np.array_equal(c.download_level(0, 0, 1), cfg1["p1_img_1"]) and np.array_equal(c.download_level(2, 0, 1), cfg1["p0_img_1"])
        with pytest.raises(KltBackendError):
            c.adopt_u8(3, clip, 320, 240 * 100000)
        with pytest.raises(KltBackendError):
            c._check(c._lib.klt_slot_adopt_u8(c._h, 3, img0.ctypes.data, 320, 240, 320))     # host memory is not adoptable
        with pytest.raises(KltBackendError):
            c._check(c._lib.klt_slot_adopt_u8(c._h, 3, clip, 300, 240, 320))                  # rows must be contiguous
        c.device_free(clip)
        assert not c.frame_resident(2) and not c.frame_resident(1)
        with pytest.raises(KltBackendError):
            c.device_free(clip)
    finally:
        c.close()


def test_the_resident_clip_example_runs():
    """examples/resident_clip.py: the cfg-5 loop on frames read in place from device memory, at a small size."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "resident_clip.py"), "--frames", "12", "--size", "640x480", "--features", "300"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "300 of 300 alive at the end" in r.stdout, r.stdout


def test_records_come_back_without_draining_the_pipeline(img0, img1, cfg1):
    """klt_featbuf_download_async / klt_download_wait: the copy is enqueued in stream order behind the tracker that wrote the records and
    lands in pinned host memory; work enqueued behind it does not disturb it; a pageable destination is refused."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(max_residue=10.0))
        c.upload(0, img0)
        c.upload(1, img1)
        c.build_pyramids_batch([0, 1])
        fl, _ = c.select(0, 100)
        c.featbuf_upload(5, fl)
        out = c.pinned_array((100,), FEAT_DTYPE)
        out["val"] = 77
        c.track_async(0, 1, 5, 6, 100)
        c.featbuf_download_async(6, out)
        c.track_async(1, 0, 6, 7, 100)                              # more work behind the copy: reads buffer 6, must not change what was copied
        c.download_wait()
        assert np.array_equal(out["val"].astype(np.int64), cfg1["trk100_r10_val"])
        ok = out["val"] >= 0
        assert np.array_equal(out["x"][ok].astype(np.float64), cfg1["trk100_r10_x"][ok])
        c.download_wait()                                           # nothing pending: returns at once
        with pytest.raises(KltBackendError):
            c.featbuf_download_async(6, np.empty(100, FEAT_DTYPE))
    finally:
        c.close()


def test_device_free_unadopts_only_frames_inside_the_freed_allocation():
    """a slot that once held an uploaded SMALL frame, then adopted a LARGER one from a klt_device_alloc buffer, holds
    no frame after klt_device_free of that buffer (its own raw buffer holds an older, smaller image: a rebuild must not read it with the
    adopted frame's size); a slot adopted from ANOTHER allocation keeps its frame and still builds."""
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4))
        small = synth.synth_pair(160, 120, seed=1)[0]
        big, other = synth.synth_pair(640, 480, seed=2)
        a, b = c.device_alloc(big.nbytes), c.device_alloc(other.nbytes)
        c.device_write(a, big)
        c.device_write(b, other)
        c.upload(0, small)                                     # slot 0 owns a 160 x 120 raw buffer
        c.build_pyramids(0)
        c.adopt_u8(0, a, 640, 480)
        c.adopt_u8(1, b, 640, 480)
        c.build_pyramids_batch([0, 1], sync=True)
        want = c.download_level(1, 0, 0).copy()
        c.device_free(a)
        assert not c.frame_resident(0) and not c.pyramids_valid(0)
        with pytest.raises(KltBackendError, match="no frame"):
            c.build_pyramids(0)
        assert c.frame_resident(1), "a slot adopted from other memory lost its frame"
        c.build_pyramids(1)
        assert np.array_equal(c.download_level(1, 0, 0), want)
        c.upload(0, small)                                     # the slot is usable again
        c.build_pyramids(0)
        assert c.level_dims(0, 0) == (160, 120)
    finally:
        c.close()


def test_uploads_into_one_slot_without_a_build_in_between_stay_ordered():
    """consecutive klt_upload_u8_async calls go round-robin over the copy streams; a slot uploaded again and again
    without a build in between (each copy then lands in a raw buffer an earlier copy -- on another stream -- was written to) always
    ends up holding the LAST frame sent."""
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4))
        w, h = 1920, 1080
        frames = [np.full((h, w), 10 * k + 5, np.uint8) for k in range(7)]
        pins = []
        for f in frames:
            p = c.pinned_array((h, w))
            p[...] = f
            pins.append(p)
        ref = Context(0)
        try:
            ref.configure(make_tc(levels=2, ss=4))
            for rounds in (2, 3, 4, 5, 7):
                for k in range(rounds):
                    c.upload_async(0, pins[k])
                c.build_pyramids(0)
                ref.upload(0, frames[rounds - 1])
                ref.build_pyramids(0)
                assert np.array_equal(c.download_level(0, 0, 0), ref.download_level(0, 0, 0)), "after %d uploads" % rounds
        finally:
            ref.close()
    finally:
        c.close()


def test_feature_buffers_mapped_into_pinned_host_memory():
    """klt_featbuf_map_host: the tracker reads and writes pinned host records in place -- same records as through device buffers and two
    copies; the ordinary copies still work on a mapped buffer; pageable memory is refused; unmapping empties the buffer; the Python
    layer unmaps before it frees the pinned arrays of a list length it evicts."""
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    c = Context(0)
    try:
        c.configure(make_tc(levels=2, ss=4, max_residue=10.0))
        f0, f1 = synth.synth_pair(640, 480, seed=9)
        c.upload(0, f0)
        c.upload(1, f1)
        c.build_pyramids_batch([0, 1], sync=True)
        n = 300
        fl, _ = c.select(0, n)
        want, _ = c.track(0, 1, fl)                                   # device buffers, synchronous copies
        rin, rout = c.pinned_array((n,), FEAT_DTYPE), c.pinned_array((n,), FEAT_DTYPE)
        rin[...] = fl
        rout["val"] = 77
        c._check(c._lib.klt_featbuf_map_host(c._h, 40, rin.ctypes.data, n))
        c._check(c._lib.klt_featbuf_map_host(c._h, 41, rout.ctypes.data, n))
        c.track_async(0, 1, 40, 41, n)
        c.sync()
        assert rout.tobytes() == want.tobytes(), "records written in place differ from the copied ones"
        assert c.featbuf_download(41, n).tobytes() == want.tobytes()            # an ordinary download of a mapped buffer
        c.featbuf_upload(40, want)                                              # ... and an upload into one: lands in the host array
        assert rin.tobytes() == want.tobytes()
        with pytest.raises(KltBackendError, match="pinned"):
            c._check(c._lib.klt_featbuf_map_host(c._h, 42, np.zeros(n, FEAT_DTYPE).ctypes.data, n))
        c._check(c._lib.klt_featbuf_map_host(c._h, 41, None, 0))                # unmapped: empty
        with pytest.raises(KltBackendError):
            c.featbuf_download(41, n)
        c.track_async(0, 1, 40, 41, n)                                          # ... and usable as an ordinary device buffer again
        assert c.featbuf_download(41, n)["val"].tolist() == c.track(0, 1, want)[0]["val"].tolist()
        # the API's own mapping: ONE pair of pinned arrays serves every list length (views), so changing the length does not remap
        for k in range(9):
            m = 50 + k
            c.host_records(m)[0][...] = fl[:m]
            c.track_enqueue(0, 1, m)
            got = c.track_complete(m)
            assert got.tobytes() == want[:m].tobytes(), m
        first_map = c._mapped_records
        assert first_map is not None and len(c._host_records[0]) >= 58
        # ... it grows with the longest list (unmapped, freed, allocated anew, mapped again) ...
        assert n > len(c._host_records[0])
        c.host_records(n)[0][...] = fl
        c.track_enqueue(0, 1, n)
        assert c.track_complete(n).tobytes() == want.tobytes()
        grown = c._mapped_records
        assert grown != first_map and len(c._host_records[0]) >= n
        # ... and a script that alternates between two list lengths keeps that one mapping (two device-wide waits per call)
        for k in range(6):
            m = (57, n)[k % 2]
            c.host_records(m)[0][...] = fl[:m]
            c.track_enqueue(0, 1, m)
            assert c.track_complete(m).tobytes() == want[:m].tobytes(), m
            assert c._mapped_records is grown
    finally:
        c.close()


@pytest.mark.gpu
def test_frames_whose_planes_would_pass_2_gb_are_refused():
    """The kernels address a plane with 32-bit byte offsets below 2 GB (raw buffer operations): a frame of 2^28 pixels or more is an
    argument error at the boundary, before anything is read or allocated; the next size down the ABI's own limits allow is not."""
    import ctypes
    from pyfeaturetrack_amd.backend import Context, KltBackendError
    ctx = Context()
    small = np.zeros((8, 8), np.uint8)                          # never read: the geometry is checked first
    for ncols, nrows in ((16384, 16384), (32768, 16384), (65535, 65535), (23171, 23171)):
        with pytest.raises(KltBackendError, match="frame too large"):
            ctx._check(ctx._lib.klt_upload_u8(ctx._h, 0, small.ctypes.data, ncols, nrows, ncols))
    taps = np.array([0.1, 0.2, 0.4, 0.2, 0.1])
    dst = np.zeros(64, np.float32)
    with pytest.raises(KltBackendError, match="bad image geometry"):
        ctx._check(ctx._lib.klt_smooth_f32(ctx._h, dst.ctypes.data, 32768, 16384, taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 5, dst.ctypes.data))
    img = (np.arange(64 * 48, dtype=np.uint32) % 251).astype(np.uint8).reshape(48, 64)
    ctx.upload(0, img)                                          # the context is still usable
    ctx.sync()
    ctx.close()
