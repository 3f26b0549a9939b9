"""The reference-shaped API under threads (INTEGRATION.md section A "Threads"): one device context per host thread, a tracking context stays
with the context it was first used on, two threads sharing one tracking context are served call by call, finalizers never wait for a
lock.  (Folded by component from the round-4 / 5 files in round 6: the tests are unchanged.)"""
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


_ROUNDS = 50


def _frames_of(k, rounds):
    w, h = _CASES[k]["size"]
    base = synth.synth_base(w, h, 40 + k)
    return [synth.synth_frame(w, h, 40 + k, r, shift=(1.7, -1.1), base=base) for r in range(rounds + 1)]


def _api_rounds(k, frames, rounds, tc=None, out=None):
    """select on frame r, track r -> r+1, replace the lost ones on r+1: the three public calls, `rounds` times"""
    sgf, trk = _api_modules()
    tc = tc or make_tc(**_CASES[k]["tc"])
    out = [] if out is None else out
    for r in range(rounds):
        fl = sgf.KLTSelectGoodFeatures(tc, frames[r], _CASES[k]["n"])
        sel = _records(fl)
        trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
        tracked = _records(fl)
        sgf.KLTReplaceLostFeatures(tc, frames[r + 1], fl)
        out.append((sel, tracked, _records(fl)))
    return out


def test_public_api_from_four_threads_at_once():
    """KLTSelectGoodFeatures -> KLTTrackFeatures -> KLTReplaceLostFeatures from four threads, each with its own
    KLT_TrackingContext (different window / levels / frame size; two with the same feature count), 50 rounds concurrently, give the
    lists the same calls give on one thread.  The reference's state is per tracking context (klt.py:43-81); here every thread gets
    its own device context (backend.default_context) and every call holds that context's lock."""
    frames = [_frames_of(k, _ROUNDS) for k in range(len(_CASES))]
    want = [_api_rounds(k, frames[k], 6) for k in range(len(_CASES))]           # single thread (the main thread's context)
    got, errors = [[] for _ in _CASES], []
    gate = threading.Barrier(len(_CASES))

    def work(k):
        try:
            gate.wait(60)
            _api_rounds(k, frames[k], _ROUNDS, out=got[k])
        except BaseException as e:              # noqa: BLE001 -- re-raised by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(_CASES))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
        assert not t.is_alive()
    if errors:
        raise errors[0]
    for k in range(len(_CASES)):
        assert len(got[k]) == _ROUNDS
        assert got[k][:6] == want[k], "thread %d differs from the single-thread run" % k
        assert any(v >= 0 for _, _, v in got[k][-1][2])


def test_threads_have_their_own_device_context_and_a_tc_stays_with_its_first():
    from pyfeaturetrack_amd.backend import context_of, default_context
    sgf, trk = _api_modules()
    main_ctx = default_context()
    assert default_context() is main_ctx
    seen = {}

    def other():
        seen["ctx"] = default_context()
        seen["again"] = default_context()
        tc = make_tc(**_CASES[0]["tc"])
        f = _frames_of(0, 1)
        sgf.KLTSelectGoodFeatures(tc, f[0], 50)
        seen["tc"], seen["tc_ctx"] = tc, context_of(tc)

    t = threading.Thread(target=other)
    t.start()
    t.join(120)
    assert seen["ctx"] is seen["again"] and seen["ctx"] is not main_ctx
    assert seen["tc_ctx"] is seen["ctx"]
    assert context_of(seen["tc"]) is seen["ctx"], "a tracking context stays with the device context it was first used on"
    # ... and goes on working from this thread (its frames and pyramids live in that context's slots)
    f = _frames_of(0, 1)
    fl = sgf.KLTSelectGoodFeatures(seen["tc"], f[0], 50)
    trk.KLTTrackFeatures(seen["tc"], f[0], f[1], fl)
    ref_tc = make_tc(**_CASES[0]["tc"])
    fl2 = sgf.KLTSelectGoodFeatures(ref_tc, f[0], 50)
    trk.KLTTrackFeatures(ref_tc, f[0], f[1], fl2)
    assert _records(fl) == _records(fl2)


def test_two_threads_sharing_one_tracking_context():
    """Two threads calling the public API on ONE KLT_TrackingContext are served one call at a time (the lock of the device context the
    tracking context is bound to): every call's result is the single-thread result for its inputs."""
    sgf, trk = _api_modules()
    k = 1
    frames = _frames_of(k, 8)
    tc = make_tc(**_CASES[k]["tc"])
    n = _CASES[k]["n"]
    want = {}
    for r in range(8):
        fl = sgf.KLTSelectGoodFeatures(tc, frames[r], n)
        sel = _records(fl)
        trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
        want[r] = (sel, _records(fl))
    got, errors = {}, []

    def work(rs):
        try:
            for _ in range(5):
                for r in rs:
                    fl = sgf.KLTSelectGoodFeatures(tc, frames[r], n)
                    sel = _records(fl)
                    trk.KLTTrackFeatures(tc, frames[r], frames[r + 1], fl)
                    got.setdefault(r, []).append((sel, _records(fl)))
        except BaseException as e:              # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(rs,)) for rs in ((0, 2, 4, 6), (1, 3, 5, 7))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
        assert not t.is_alive()
    if errors:
        raise errors[0]
    for r in range(8):
        assert len(got[r]) == 5 and all(g == want[r] for g in got[r]), "round %d" % r


def test_finalizers_never_wait_for_a_context_somebody_else_is_inside():
    """A tracking context that dies while another thread holds its device context's lock leaves its release for the next holder
    (Context._when_free / settle_deferred) instead of blocking the thread the collector happens to run on."""
    import gc
    from pyfeaturetrack_amd.backend import context_of
    sgf, _ = _api_modules()
    f = _frames_of(0, 1)
    tc = make_tc(**_CASES[0]["tc"])
    sgf.KLTSelectGoodFeatures(tc, f[0], 30)
    ctx = context_of(tc)
    base = tc._klt_slots[0]
    held, release = threading.Event(), threading.Event()

    def holder():
        with ctx.lock:
            held.set()
            release.wait(60)

    t = threading.Thread(target=holder)
    t.start()
    assert held.wait(60)
    del tc
    gc.collect()                                    # the finalizer runs here and must not block
    assert len(ctx._deferred) == 1
    release.set()
    t.join(60)
    tc2 = make_tc(**_CASES[0]["tc"])
    sgf.KLTSelectGoodFeatures(tc2, f[0], 30)        # the next call settles the deferred release and reuses the slots
    assert not ctx._deferred and tc2._klt_slots[0] == base


def test_one_context_per_host_thread():
    """include/klt_gpu.h: "One context per host thread / device; no shared mutable globals" (ctypes releases the GIL during every call).
    Six threads, a context each, run upload + pyramids + selection + tracking + replacement on their own frames at the same time, eight
    rounds each with a different frame size per thread (so buffers are grown, LDS attributes set and kernels loaded concurrently); every
    round gives exactly what the same calls give on one thread."""
    import threading
    from pyfeaturetrack_amd.backend import Context
    sizes = [(320, 240, 150), (648, 486, 400), (500, 380, 300), (1280, 720, 1500), (402, 302, 200), (960, 540, 900)]
    ROUNDS = 8
    frames = [[synth.synth_pair(w, h, 100 * k + r) for r in range(ROUNDS)] for k, (w, h, _) in enumerate(sizes)]

    def work(k, out, rounds):
        n = sizes[k][2]
        try:
            c = Context(0)
            try:
                c.configure(make_tc(max_residue=10.0, levels=3 if k % 2 else 2, ss=2 if k % 2 else 4))
                for r in range(rounds):
                    f0, f1 = frames[k][r]
                    c.upload(0, f0)
                    c.upload(1, f1)
                    c.build_pyramids_batch([0, 1], sync=False)
                    fl, placed = c.select(0, n)
                    trk, _ = c.track(0, 1, fl)
                    rep, _ = c.select(1, n, mode=2, fl=trk)
                    out.append((placed, fl.tobytes(), trk.tobytes(), rep.tobytes()))
            finally:
                c.close()
        except BaseException as e:          # noqa: BLE001  (re-raised by the main thread)
            out.append(e)

    want = [[] for _ in sizes]
    for k in range(len(sizes)):
        work(k, want[k], 3)
    got = [[] for _ in sizes]
    threads = [threading.Thread(target=work, args=(k, got[k], ROUNDS)) for k in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    for k in range(len(sizes)):
        for e in got[k] + want[k]:
            if isinstance(e, BaseException):
                raise e
        assert len(got[k]) == ROUNDS and got[k][:3] == want[k], "thread %d" % k
        assert any(rec[0] > 0 for rec in got[k])
