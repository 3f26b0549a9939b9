"""BASELINE cfg-4 and cfg-5 at the counts BASELINE.json states: 256 independent 1280x720 pairs in one batched pass, and a 512-frame
3840x2160 sequence with 20000 features and replacement after every frame.  The frame sizes and feature counts are
pinned against the reference's own outputs elsewhere (tests/golden/baseline_sizes.npz); these tests cover what only the counts
exercise: launches over hundreds of frames, the slot ring, score sets, descriptor tables and the tracker enqueued ahead of the
selection's outcome over hundreds of steps."""
import hashlib
import os
import time

import numpy as np
import pytest

from helpers import make_tc, params_from_tc
from pyfeaturetrack_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ko():
    from oracle import klt_oracle
    return klt_oracle


def _cores():
    import bench
    return bench.usable_cores()


def _same_records(got, want, what):
    for name in ("val", "x", "y"):
        bad = np.flatnonzero(got[name] != want[name])
        assert bad.size == 0, "%s: %d records differ in %s; first at %d: %r vs %r" % (what, bad.size, name, bad[0], got[name][bad[0]],
                                                                                    want[name][bad[0]])


def _digest(rec):
    cols = np.stack([rec["x"].view(np.int32), rec["y"].view(np.int32), rec["val"].astype(np.int32)], axis=-1)
    return hashlib.sha256(np.ascontiguousarray(cols).tobytes()).hexdigest()[:16]


# ================================================================================================ cfg-4: 256 pairs
def test_cfg4_256_pairs_in_one_batched_pass(ko):
    """256 independent 1280x720 pairs (seeds 0..255) resident at once, 2000 features each, 7x7, 3 levels / ss 4: ONE
    klt_build_pyramids_batch_async call for the 512 frames, one selection per pair, ONE klt_track_batch_async call for the 256 lists
    into a device-side [256 x 2000] table -- and the WHOLE table against the oracle's records (all host cores), plus the oracle's own
    selection for every 16th pair.  Every pair recovers the imposed shift."""
    from concurrent.futures import ThreadPoolExecutor
    from pyfeaturetrack_amd.backend import Context
    W, H, NF, NP = 1280, 720, 2000, 256
    tc = make_tc(levels=3, ss=4)
    p = params_from_tc(tc)
    with ThreadPoolExecutor(max_workers=_cores()) as ex:
        frames = list(ex.map(lambda i: synth.synth_pair(W, H, seed=i), range(NP)))
    c = Context(0)
    try:
        c.configure(tc)
        for k, (f0, f1) in enumerate(frames):
            c.upload(2 * k, f0)
            c.upload(2 * k + 1, f1)
        T_IN, T_OUT, V_IN, V_OUT = 0, 1, 1000, 2000
        c.featbuf_alloc(T_IN, NP * NF)
        c.featbuf_alloc(T_OUT, NP * NF)
        c.build_pyramids_batch(list(range(2 * NP)))                      # one call: 512 frames share the launches
        for k in range(NP):
            c.featbuf_view(V_IN + k, T_IN, k * NF, NF)
            c.featbuf_view(V_OUT + k, T_OUT, k * NF, NF)
            c.select_async(2 * k, 1, True, V_IN + k, NF)
        c.track_batch_async([(2 * k, 2 * k + 1, V_IN + k, V_OUT + k) for k in range(NP)], NF)      # one call: 256 lists
        fl_in = c.featbuf_download(T_IN, NP * NF).reshape(NP, NF)
        out = c.featbuf_download(T_OUT, NP * NF).reshape(NP, NF)
        # ... and once more, into the same table: a second step of the bench must not change a record
        c.build_pyramids_batch(list(range(2 * NP)))
        c.track_batch_async([(2 * k, 2 * k + 1, V_IN + k, V_OUT + k) for k in range(NP)], NF)
        assert c.featbuf_download(T_OUT, NP * NF).tobytes() == out.tobytes()
    finally:
        c.close()
    assert (fl_in["val"] > 0).all(), "a selection left slots empty"
    import bench
    n = _cores()
    for k in range(NP):
        want = bench.oracle_track(ko, p, frames[k][0], frames[k][1], fl_in[k], threads=n)
        _same_records(out[k], want, "cfg-4 pair %d (seed %d)" % (k, k))
        live = out[k]["val"] == 0
        assert live.sum() > 0.9 * NF, (k, int(live.sum()))
        dx, dy = np.median(out[k]["x"][live] - fl_in[k]["x"][live]), np.median(out[k]["y"][live] - fl_in[k]["y"][live])
        assert abs(dx - 3.3) < 0.05 and abs(dy + 2.1) < 0.05, (k, dx, dy)
    ko.set_threads(n)
    try:
        for k in range(0, NP, 16):
            osel = ko.select_good_features(p, frames[k][0].astype(np.float32), NF)
            _same_records(fl_in[k], osel, "cfg-4 selection of pair %d" % k)
    finally:
        ko.set_threads(1)


# ================================================================================================ cfg-5: 512 frames
W5, H5, NF5, NFRAMES5 = 3840, 2160, 20000, 512


def _cfg5_tc():
    tc = make_tc(levels=3, ss=4, max_residue=10.0)
    tc.sequentialMode = True
    return tc


def test_cfg5_512_frames_with_replacement_every_frame(ko):
    """One 3840x2160 sequence of 512 frames (the periodic texture moved by (3.3, -2.1) per frame), 20000 features, sequential mode,
    KLTReplaceLostFeatures after every frame -- twice: through KLTTrackSequence (frames from host memory, ring of three slots, the
    device-side [frames x features] table) and through the per-frame ABI loop bench.py --config cfg5 times (build stream, prepared
    scores, the next tracker enqueued before the host looks at the selection).  Checked:
      * the list after EVERY frame is the same in both (digests of all 512 rows);
      * the oracle's chain (track + REPLACING_SOME selection) for the first 16 frames, and again for the last 8 starting from the
        device's list at frame 503 -- the records, bit for bit;
      * every frame recovers the imposed shift within 0.01 px, 20000 features are alive at the end;
      * device memory is flat from frame 64 on, apart from the table's own chunks (slots, score sets, descriptor tables and events
        are recycled)."""
    from pyfeaturetrack_amd.backend import REPLACING_SOME, SELECTING_ALL, default_context
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    tc = _cfg5_tc()
    p = params_from_tc(tc)
    phases = synth.sequence_phases(W5, H5, 4, workers=_cores())

    def sequence(start=0):
        return synth.periodic_sequence(W5, H5, 4, NFRAMES5, phases=phases, start=start)

    ctx = default_context()
    free_at = {}

    def frames_watched():
        for k, f in enumerate(sequence()):
            if k % 64 == 0 or k == NFRAMES5 - 1:
                free_at[k] = ctx.device_memory()[0]
            yield f

    t0 = time.perf_counter()
    ft = KLTTrackSequence(tc, frames_watched(), NF5, replace_lost=True)
    t_seq = time.perf_counter() - t0
    assert ft.rec.shape == (NFRAMES5, NF5)
    rows = ft.rec

    # ---- every frame: all alive after the replacement, and the survivors moved by the imposed shift
    for k in range(1, NFRAMES5):
        assert (rows[k]["val"] >= 0).all(), "frame %d: %d slots dead after the replacement" % (k, int((rows[k]["val"] < 0).sum()))
        kept = rows[k]["val"] == 0                                  # tracked (a replaced slot carries its eigenvalue)
        assert kept.sum() > 0.98 * NF5, (k, int(kept.sum()))
        dx = np.median(rows[k]["x"][kept] - rows[k - 1]["x"][kept])
        dy = np.median(rows[k]["y"][kept] - rows[k - 1]["y"][kept])
        assert abs(dx - 3.3) <= 0.01 and abs(dy + 2.1) <= 0.01, (k, dx, dy)

    # ---- memory: the table grows by one chunk of 64 rows per 64 frames, nothing else does
    chunk = 64 * NF5 * 16                                           # 20.5 MB; one 4K slot is 115 MB, one score set 66 MB
    marks = sorted(k for k in free_at if k >= 64)
    for a, b in zip(marks, marks[1:]):
        grown = free_at[a] - free_at[b]
        assert grown <= 2 * chunk + (4 << 20), "free device memory fell by %.1f MB between frames %d and %d" % (grown / 1e6, a, b)

    # ---- the per-frame ABI loop of bench.py --config cfg5, on a ring of three slots with the frames uploaded as they come
    S = [100, 101, 102]
    FB = [7000, 7001]
    digests = [None] * NFRAMES5
    ctx.set_option(15, 1)                                           # KLT_OPT_BUILD_STREAM
    redone = 0
    try:
        gen = sequence()
        ctx.upload(S[0], next(gen))
        ctx.build_pyramids(S[0], sync=False)
        ctx.select_async(S[0], SELECTING_ALL, True, FB[0], NF5)
        digests[0] = _digest(ctx.featbuf_download(FB[0], NF5))
        snapshot = None

        def stage(k, frame):
            ctx.upload(S[k % 3], frame)
            ctx.build_pyramids(S[k % 3], sync=False)
            ctx.select_prepare(S[k % 3])

        def track(k):
            ctx.track_async(S[(k - 1) % 3], S[k % 3], FB[(k - 1) % 2], FB[k % 2], NF5)

        free_loop = {}
        nxt = next(gen)
        stage(1, nxt)
        track(1)
        for k in range(1, NFRAMES5):
            nxt = next(gen, None)
            ctx.select_begin(S[k % 3], REPLACING_SOME, True, FB[k % 2], NF5)
            if nxt is not None:
                stage(k + 1, nxt)
                track(k + 1)
            if ctx.select_finish() and nxt is not None:           # the selection rewrote the list after the tracker had read it:
                redone += 1                                       # the tracker goes out once more (part of the protocol)
                track(k + 1)
            rec = ctx.featbuf_download(FB[k % 2], NF5)             # (the look is the test's: the tracker of k + 1 only READS this list)
            digests[k] = _digest(rec)
            if k == NFRAMES5 - 9:
                snapshot = rec.copy()
            if k % 64 == 0:
                free_loop[k] = ctx.device_memory()[0]
    finally:
        ctx.select_finish()
        ctx.set_option(15, 0)
        for s in S:
            ctx.slot_free(s)
    want = [_digest(rows[k]) for k in range(NFRAMES5)]
    differ = [k for k in range(NFRAMES5) if digests[k] != want[k]]
    assert not differ, "KLTTrackSequence and the per-frame ABI loop differ first at frame %d (%d frames in all)" % (differ[0], len(differ))
    marks = sorted(free_loop)
    assert max(free_loop[k] for k in marks[1:]) - min(free_loop[k] for k in marks[1:]) <= (4 << 20), free_loop     # flat from frame 128 on

    # ---- the oracle: the first 16 frames from its own selection, the last 8 from the device's list at frame 503
    n = _cores()
    ko.set_threads(n)
    try:
        gen = sequence()
        f_prev = next(gen).astype(np.float32)
        ofl = ko.select_good_features(p, f_prev, NF5)
        _same_records(rows[0], ofl, "cfg-5 initial selection")
        P_prev = ko.Pyramids(p, f_prev)
        for k in range(1, 17):
            f = next(gen).astype(np.float32)
            P_cur = ko.Pyramids(p, f)
            ko.track_features(p, P_prev, P_cur, ofl)
            ofl = ko.select_good_features(p, f, NF5, mode=2, fl=ofl)
            _same_records(rows[k], ofl, "cfg-5 list after frame %d" % k)
            P_prev = P_cur
        k0 = NFRAMES5 - 9
        assert _digest(snapshot) == want[k0]
        tail = list(sequence(start=k0))
        ofl = snapshot.copy()
        P_prev = ko.Pyramids(p, tail[0].astype(np.float32))
        for j in range(1, 9):
            f = tail[j].astype(np.float32)
            P_cur = ko.Pyramids(p, f)
            ko.track_features(p, P_prev, P_cur, ofl)
            ofl = ko.select_good_features(p, f, NF5, mode=2, fl=ofl)
            _same_records(rows[k0 + j], ofl, "cfg-5 list after frame %d (from the snapshot at %d)" % (k0 + j, k0))
            P_prev = P_cur
    finally:
        ko.set_threads(1)
    print("cfg-5: KLTTrackSequence over %d frames from host memory: %.3f ms per frame; trackers repeated in the ABI loop: %d"
          % (NFRAMES5, t_seq / (NFRAMES5 - 1) * 1e3, redone))
