"""Round-3 GPU checks: cfg-3 at its own size with the affine consistency check on, the driver's bench command line, and a
star-import driver (the reference's example1.py call sequence) through pyfeaturetrack_amd/compat in a fresh process."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from helpers import params_from_tc
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.params import affine_params_from_tc

pytestmark = pytest.mark.gpu

W, H, NF = 1920, 1080, 5000
STATE_FIELDS = ("valid", "aff_x", "aff_y", "Axx", "Ayx", "Axy", "Ayy", "pad")


def cfg3_tc(mode):
    import bench
    tc = bench.cfg3_context()                      # 15x15, 4 levels / ss 2 (border 108), affine window 15x15
    tc.affineConsistencyCheck = mode
    return tc


def shifted_frames():
    import bench
    return bench.cfg3_frames(4)                    # what `bench.py --config cfg3` times: pure translation by (1.1, -0.7) per frame


def warped_frames():
    """a small similarity + shear per frame on top of the translation: the affine matrices have something to converge to"""
    base = synth.synth_base(W, H, 1)
    A_step = np.array([[1.0015, 0.0012], [-0.0009, 0.9988]])
    frames, A = [], np.eye(2)
    for k in range(4):
        frames.append(synth.warp_frame(base, A, (1.1 * k, -0.7 * k)))
        A = A_step @ A
    return frames


def three_calls_gpu(frames, tc):
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        c.configure(tc)
        for k, f in enumerate(frames):
            c.upload(k, f)
        c.build_pyramids_batch(list(range(len(frames))), sync=True)
        fl, placed = c.select(0, NF, use_pyramid=True)
        assert placed == NF
        c.affine_alloc(0, NF)
        c.featbuf_upload(0, fl)
        hist = [(fl.copy(), None)]
        for k in range(1, len(frames)):
            c.track_affine_async(k - 1, k, k - 1, k, NF, 0)          # through the asynchronous ABI entry point bench.py times
            hist.append((c.featbuf_download(k, NF), c.affine_download(0, NF)))
        return hist
    finally:
        c.close()


def three_calls_oracle(frames, tc):
    from oracle import klt_oracle as ko
    p, ap = params_from_tc(tc), affine_params_from_tc(tc)
    ko.set_threads(min(16, os.cpu_count() or 1))
    try:
        f32 = [f.astype(np.float32) for f in frames]
        fl = ko.select_good_features(p, f32[0], NF)
        st = ko.AffineState(ap, NF)
        P = [ko.Pyramids(p, f) for f in f32]
        hist = [(fl.copy(), None)]
        for k in range(1, len(frames)):
            ko.track_features_affine(p, P[k - 1], P[k], fl, st)
            hist.append((fl.copy(), st.rec.copy()))
        return hist
    finally:
        ko.set_threads(1)


def assert_same_history(g, o, what):
    for k, ((gfl, grec), (ofl, orec)) in enumerate(zip(g, o)):
        assert np.array_equal(gfl["val"], ofl["val"]), "%s, call %d: %d status codes differ" % (what, k, int((gfl["val"] != ofl["val"]).sum()))
        assert np.array_equal(gfl["x"], ofl["x"]) and np.array_equal(gfl["y"], ofl["y"]), "%s, call %d: positions" % (what, k)
        if grec is not None:
            for name in STATE_FIELDS:
                assert np.array_equal(grec[name], orec[name]), "%s, call %d: affine state field %s" % (what, k, name)


@pytest.mark.parametrize("mode,frames_of", [(2, "shifted"), (2, "warped"), (1, "warped"), (0, "warped")])
def test_cfg3_full_size_with_the_affine_check_on(mode, frames_of):
    """BASELINE cfg-3 at its real geometry -- 1920x1080, 15x15 window, 4 levels / ss 2, 5000 features, four frames = three
    KLTTrackFeatures calls (the first stores the templates, the second and third run the check; interface:
    /root/reference trackFeatures.py:347-399) -- HIP == oracle on val, x, y and on valid, aff_x / aff_y, the four entries of A and
    the iteration count of every feature, after every call.  Parity of the check itself is UNPINNED (the reference does not define
    the functions it calls there); this is the implementation against the stated specification."""
    frames = shifted_frames() if frames_of == "shifted" else warped_frames()
    tc = cfg3_tc(mode)
    g, o = three_calls_gpu(frames, tc), three_calls_oracle(frames, tc)
    assert_same_history(g, o, "mode %d, %s frames" % (mode, frames_of))
    last_fl, last_rec = g[-1]
    assert (last_fl["val"] == 0).sum() > 0.9 * NF
    live = last_fl["val"] == 0
    assert last_rec["pad"][live].min() >= 1, "the check ran on every surviving feature"
    if frames_of == "warped" and mode == 2:
        want = np.linalg.matrix_power(np.array([[1.0015, 0.0012], [-0.0009, 0.9988]]), 3)
        got = np.array([[np.median(last_rec["Axx"][live]), np.median(last_rec["Axy"][live])],
                        [np.median(last_rec["Ayx"][live]), np.median(last_rec["Ayy"][live])]])
        assert np.abs(got - want).max() < 4e-3, (got, want)


def run_bench(extra_args, timeout=1200, **env_kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE")}
    env.update(env_kw)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_the_drivers_command_line_prints_a_clean_record():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- exactly what the driver runs at round end: no figure above its roof,
    iteration counters that describe the launches they are divided by, 72 distinct pairs all checked against the oracle, a timed
    phase long enough to be seen from outside the process, and a value that does not depend on K."""
    line = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    roof = line["roofline"]
    assert line["parity_checked"] is True and line["parity_cases"] == 72 and line["max_abs_dx"] <= 1e-3
    assert line["config"]["resident_pairs"] == 72 and line["config"]["pairs_per_step"] == 72 and line["config"]["contexts"] == 3
    assert 0 < roof["frac"] < 1 and 0 < roof["frac_moved"] < roof["frac"] and 0 < roof["step_frac"] < 1
    assert roof["kernel"] == "smooth_grad_l0" and roof["launch_us_source"] == "dispatch timestamps"
    for name, k in roof["kernels"].items():
        assert 0 <= k["frac"] < 1 and k["GBps"] < roof["peak"], name
    for l, it in enumerate(roof["newton_iterations_per_level"]):
        assert 5000 <= it <= 25000, "level %d: %.0f Newton iterations per pair" % (l, it)
    assert abs(roof["step_algorithmic_bytes"] / roof["step_algorithmic_bytes_formula"] - 1) < 1e-6
    assert line["extra"]["region_ms_per_step"]["timed_s_total"] >= 2.0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1
    # the same figures from a run with five times the steps per region
    long = run_bench(["--gpus", "1", "--steps", "100", "--warmup", "5", "--repeats", "8", "--no-cpu-baseline", "--no-extras"])
    assert abs(long["roofline"]["step_algorithmic_bytes"] / roof["step_algorithmic_bytes"] - 1) < 0.05
    assert abs(long["value"] / line["value"] - 1) < 0.08, (long["value"], line["value"])     # (seen: 0.3-1.7 %)


def test_bench_refuses_figures_above_the_roof():
    import bench
    ok = {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 0.2, "kernels": {"track": {"GBps": 900.0, "frac": 0.11}}}}
    assert bench.check_fractions(ok) == []
    for path, bad in (("step_frac", {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 2.6}}),
                      ("GBps", {"roofline": {"peak": 8000.0, "kernels": {"track": {"GBps": 72833.0, "frac": 0.5}}}}),
                      ("frac_moved", {"roofline": {"frac_moved": 1.01}})):
        assert any(path in m for m in bench.check_fractions(bad)), path
    fd = os.open(os.devnull, os.O_WRONLY)
    try:
        with pytest.raises(SystemExit):
            bench.emit(fd, {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 2.6}})
    finally:
        os.close(fd)


@pytest.mark.parametrize("cfg,extra", [("cfg1", []), ("cfg3", ["--steps", "10", "--repeats", "5"])])
def test_every_config_line_carries_checker_roofline_and_baseline(cfg, extra):
    line = run_bench(["--config", cfg] + extra)
    assert line["parity_checked"] is True, line
    assert 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["step_frac"] < 1
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    if cfg == "cfg3":
        k = line["roofline"]["kernels"]
        assert k["affine_check"]["timed_by"] == "dispatch timestamps" and k["track"]["timed_by"] == "dispatch timestamps"
        assert line["roofline"]["affine_iterations_per_checked_feature"] >= 1


# a driver written the way scripts written against the reference are: star imports of the reference's module names, time.clock()
STAR_IMPORT_DRIVER = '''
from __future__ import print_function
from klt import *
from PIL import Image
from selectGoodFeatures import *
from writeFeatures import *
from trackFeatures import *
import time

tc = KLT_TrackingContext()
tc.nSkippedPixels = 0
tc.max_residue = 10.0
KLTPrintTrackingContext(tc)
first, second = Image.open("img0.pgm"), Image.open("img1.pgm")
features = KLTSelectGoodFeatures(tc, first, 50)
for k, f in enumerate(features):
    print("Feature #{0}:  ({1},{2}) with value of {3}".format(k, f.x, f.y, f.val))
KLTWriteFeatureListToPPM(features, first, "feat1.ppm")
calls, started = 0, time.clock()
for _ in range(100):
    KLTTrackFeatures(tc, first, second, features)
    KLTTrackFeatures(tc, second, first, features)
    calls += 2
print("seconds per call", (time.clock() - started) / calls)
print("remaining", KLTCountRemainingFeatures(features))
for k, f in enumerate(features):
    print("Feature #{0}:  ({1},{2}) with value of {3}".format(k, f.x, f.y, f.val))
KLTWriteFeatureListToPPM(features, second, "feat2.ppm")
'''


def test_star_import_driver_through_compat(tmp_path, golden_dir):
    """north_star: "example1.py runs unchanged".  A script that imports the reference's top-level module names with `import *` and
    times itself with time.clock() (/root/reference example1.py:10-14, :17-65 -- the call sequence, not the file) runs in a fresh
    process with only PYTHONPATH pointing at this repository and its compat directory; the two PPM files and the list after the
    200-call ping-pong are the reference's own (tests/golden/example1.npz)."""
    pytest.importorskip("PIL.Image")
    import shutil
    for name in ("img0.pgm", "img1.pgm"):
        shutil.copy(os.path.join(golden_dir, name), tmp_path / name)
    script = tmp_path / "driver.py"
    script.write_text(STAR_IMPORT_DRIVER)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, "pyfeaturetrack_amd", "compat")]))
    r = subprocess.run([sys.executable, str(script)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ex = np.load(os.path.join(golden_dir, "example1.npz"))
    for name in ("feat1", "feat2"):
        data = (tmp_path / (name + ".ppm")).read_bytes()
        assert np.array_equal(np.frombuffer(hashlib.sha256(data).digest(), np.uint8), ex[name + "_ppm_sha"]), name
    feats = [l for l in r.stdout.splitlines() if l.startswith("Feature #")]
    assert len(feats) == 100
    for k in range(50):
        x, y, v = ex["pp_after_200_x"][k], ex["pp_after_200_y"][k], int(ex["pp_after_200_val"][k])
        assert feats[50 + k] == "Feature #{0}:  ({1},{2}) with value of {3}".format(k, float(x), float(y), v), k
    assert "remaining %d" % int((ex["pp_after_200_val"] >= 0).sum()) in r.stdout


# ------------------------------------------------------------------------------------------------ Python API: resident frames
def _api_modules():
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import trackFeatures as trk
    sgf.KLT_verbose = trk.KLT_verbose = 0
    return sgf, trk


def _records(fl):
    return [(f.x, f.y, f.val) for f in fl]


@pytest.mark.skipif(bool(os.environ.get("KLT_NO_FRAME_CACHE")), reason="the frame cache was switched off through the environment")
def test_python_api_keeps_frames_resident_and_notices_changes():
    """The reference-shaped API with the frame cache (_frames.py): the ping-pong of example1 uploads and builds nothing after its
    first round trip, results are those of a cache-less run, an image edited in place is seen as new, KLTForgetFrames voids the
    cache, and KLT_NO_FRAME_CACHE=1 (a fresh process) gives the same lists."""
    from helpers import make_tc
    from pyfeaturetrack_amd.backend import default_context
    from pyfeaturetrack_amd._frames import cache_of
    sgf, trk = _api_modules()
    try:
        base = synth.synth_base(640, 480, 21)
        f = [synth.synth_frame(640, 480, 21, k, shift=(1.7, -1.1), base=base) for k in range(3)]
        n = 400

        def run(tc, forget):
            out = []
            fl = sgf.KLTSelectGoodFeatures(tc, f[0], n)
            out.append(_records(fl))
            for k in range(6):                                   # ping-pong between two frames, then a third one
                a, b = (f[0], f[1]) if k % 2 == 0 else (f[1], f[0])
                if forget:
                    trk.KLTForgetFrames(tc)
                trk.KLTTrackFeatures(tc, a, b, fl)
                out.append(_records(fl))
            trk.KLTTrackFeatures(tc, f[0], f[2], fl)
            out.append(_records(fl))
            sgf.KLTReplaceLostFeatures(tc, f[2], fl)             # the image frame 2's slot holds: selected on level 0 of its pyramid
            out.append(_records(fl))
            return out

        tc1, tc2 = make_tc(levels=2, ss=4, max_residue=10.0), make_tc(levels=2, ss=4, max_residue=10.0)
        want = run(tc1, forget=True)
        ctx = default_context()
        ctx.timing_enable(1)
        got = run(tc2, forget=False)
        launches = {k["name"]: k["launches"] for k in ctx.timing_read()}
        ctx.timing_enable(0)
        assert got == want
        # select(f0) builds f0's pyramid, track #1 builds f1's, track #7 builds f2's: three level-0 launches in all
        assert launches.get("smooth_grad_l0") == 3, launches
        assert launches.get("track") == 7
        # an in-place edit of the frame is a new frame
        tc3 = make_tc(levels=2, ss=4, max_residue=10.0)
        g0, g1 = f[0].copy(), f[1].copy()
        fl = sgf.KLTSelectGoodFeatures(tc3, g0, n)
        trk.KLTTrackFeatures(tc3, g0, g1, fl)
        first = _records(fl)
        g1[:] = f[2]                                             # same object, other pixels
        fl2 = sgf.KLTSelectGoodFeatures(tc3, g0, n)
        trk.KLTTrackFeatures(tc3, g0, g1, fl2)
        tc4 = make_tc(levels=2, ss=4, max_residue=10.0)
        fl3 = sgf.KLTSelectGoodFeatures(tc4, f[0], n)
        trk.KLTTrackFeatures(tc4, f[0], f[2], fl3)
        assert _records(fl2) == _records(fl3) and _records(fl2) != first
        assert len(cache_of(tc3).held) == 2
        # sequential mode keeps working through the cache (frame 2 becomes frame 1)
        tc5, tc6 = make_tc(levels=2, ss=4), make_tc(levels=2, ss=4)
        tc5.sequentialMode = True
        a = sgf.KLTSelectGoodFeatures(tc5, f[0], n)
        b = sgf.KLTSelectGoodFeatures(tc6, f[0], n)
        for k in (1, 2):
            trk.KLTTrackFeatures(tc5, f[k - 1], f[k], a)
            trk.KLTTrackFeatures(tc6, f[k - 1], f[k], b)
            assert _records(a) == _records(b), k
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def test_python_api_without_the_frame_cache_gives_the_same_example1(tmp_path, golden_dir):
    """KLT_NO_FRAME_CACHE=1: every call uploads and rebuilds what it is given, as the reference does; example1's files and lists
    are the same."""
    env = dict(os.environ, KLT_NO_FRAME_CACHE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "example1.py"), "--out", str(tmp_path), "--iterations", "10"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ex = np.load(os.path.join(golden_dir, "example1.npz"))
    data = (tmp_path / "feat1.ppm").read_bytes()
    assert np.array_equal(np.frombuffer(hashlib.sha256(data).digest(), np.uint8), ex["feat1_ppm_sha"])
    last = [l for l in r.stdout.splitlines() if l.startswith("Feature #49:")][-1]
    x, y, v = ex["pp_after_20_x"][49], ex["pp_after_20_y"][49], int(ex["pp_after_20_val"][49])
    assert last == "Feature #49:  ({0},{1}) with value of {2}".format(float(x), float(y), v)


# ------------------------------------------------------------------------------------------------ new ABI entry points
def test_gatherv_timeout_and_the_fixed_teardown_paths():
    """Round-3 entry points on a one-rank communicator: klt_gatherv_featbuf_async (a count per rank), klt_comm_set_timeout (a wait
    that cannot hang), klt_comm_destroy followed by a fence on a buffer that took part in a collective (ADVICE: the buffer kept an
    event of the destroyed communicator), a table regrown while its gather may still run (ADVICE: sync_all now waits for the side
    stream), klt_featbuf_upload_async and klt_affine_copy_async."""
    import ctypes as C
    from pyfeaturetrack_amd._abi import load_library
    from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE, KltBackendError
    lib = load_library()
    c = Context(0)
    try:
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.gatherv_featbuf_async(0, 1, [4], 0)
        with pytest.raises(KltBackendError, match="klt_comm_init_rank"):
            c.comm_set_timeout(10.0)
        uid = (C.c_uint8 * 128)()
        assert lib.klt_comm_unique_id(uid) == 0
        c.comm_init(1, 0, bytes(uid))
        n = 2000
        fl = np.zeros(n, FEAT_DTYPE)
        fl["x"], fl["y"], fl["val"] = np.arange(n), -np.arange(n), np.arange(n) % 5 - 2
        rin, rout = c.host_records(n)
        rin[:] = fl
        assert lib.klt_featbuf_upload_async(c._h, 0, rin.ctypes.data, n) == 0
        c.gatherv_featbuf_async(0, 1, [n], 0)
        c.comm_set_timeout(5000.0)
        c.comm_wait()                                       # completes: no timeout
        assert np.array_equal(c.featbuf_download(1, n), fl)
        with pytest.raises(KltBackendError, match="count"):
            c.gatherv_featbuf_async(0, 1, [-1], 0)
        with pytest.raises(KltBackendError, match="gatherv"):
            c.gatherv_featbuf_async(0, 1, [n], 2)
        # a gather into a table that has to grow while an earlier, smaller gather into it may still be running
        for k in range(20):
            c.gather_featbuf_async(0, 2, 100 + 90 * k, root=0)
        assert np.array_equal(c.featbuf_download(2, 1810), fl[:1810])
        # destroy, then fence a buffer that took part: no stale event is waited on
        c.comm_destroy()
        c.comm_fence_featbuf(0)
        c.comm_fence_featbuf(1)
        c.featbuf_upload(0, fl)
        assert np.array_equal(c.featbuf_download(0, n), fl)
        # a second communicator on the same context works (buffers keep no event of the first)
        assert lib.klt_comm_unique_id(uid) == 0
        c.comm_init(1, 0, bytes(uid))
        c.allgather_featbuf_async(0, 3, n)
        assert np.array_equal(c.featbuf_download(3, n), fl)
        c.comm_destroy()
        # affine state snapshot
        from helpers import make_tc
        tc = make_tc(levels=2, ss=2)
        tc.affineConsistencyCheck = 2
        c.configure(tc)
        c.affine_alloc(0, 64)
        c.affine_copy(1, 0, 64, with_templates=True)
        a, b = c.affine_download(0, 64), c.affine_download(1, 64)
        assert np.array_equal(a, b) and np.all(a["Axx"] == 1) and np.all(a["valid"] == 0)
        with pytest.raises(KltBackendError):
            c.affine_copy(2, 7, 64)
    finally:
        c.close()


def test_tracker_enqueued_before_the_selections_look_is_repeated_when_the_list_changes():
    """klt_select_finish returns 1 when the host's look made the selection rewrite the list after the first half's launches had run
    (here: a score ramp -- one dependency chain across the frame -- needs far more minimum-distance passes than a fresh context
    enqueues before it looks); a tracker launched in between has then read an unfinished list and must be launched again, which is
    what KLTTrackSequence and bench.py --config cfg5 do.  The repeated launch gives the records of the plain order (select, then
    track), whether or not the look asked for more (real scores: either way)."""
    from helpers import make_tc
    from pyfeaturetrack_amd.backend import Context, SELECTING_ALL
    n, w, h = 400, 500, 300
    base = synth.synth_base(w, h, 13)
    f0, f1 = synth.synth_frame(w, h, 13, 0, shift=(1.5, -1.0), base=base), synth.synth_frame(w, h, 13, 1, shift=(1.5, -1.0), base=base)
    tc = make_tc(levels=2, ss=2, mindist=10)
    p = params_from_tc(tc)
    bx, by = int(max(p.borderx, p.window_width / 2.0)), int(max(p.bordery, p.window_height / 2.0))
    ys, xs = np.mgrid[0:h - 2 * by, 0:w - 2 * bx]
    ramp = (10.0 + xs + 0.001 * ys).astype(np.float32)
    for scores in (ramp, None):
        c = Context(0)                                         # fresh: it enqueues its default number of passes before the first look
        try:
            c.configure(tc)
            c.upload(0, f0)
            c.upload(1, f1)
            c.build_pyramids_batch([0, 1], sync=True)
            # plain order
            if scores is not None:
                c.set_score_override(scores)
            c.select_async(0, SELECTING_ALL, True, 0, n)
            c.track_async(0, 1, 0, 1, n)
            want_list, want_trk = c.featbuf_download(0, n), c.featbuf_download(1, n)
            assert (want_list["val"] > 0).sum() > n // 2
        finally:
            c.close()
        c = Context(0)
        try:
            c.configure(tc)
            c.upload(0, f0)
            c.upload(1, f1)
            c.build_pyramids_batch([0, 1], sync=True)
            if scores is not None:
                c.set_score_override(scores)
            c.select_begin(0, SELECTING_ALL, True, 0, n)
            c.track_async(0, 1, 0, 2, n)                       # reads the list the selection may still be working on
            rewritten = c.select_finish()
            assert rewritten or scores is None, "the ramp needs more passes than a fresh context enqueues before it looks"
            if rewritten:
                c.track_async(0, 1, 0, 2, n)
            assert np.array_equal(c.featbuf_download(0, n), want_list)
            assert np.array_equal(c.featbuf_download(2, n), want_trk)
        finally:
            c.close()


def test_frame_cache_on_random_call_sequences():
    """Random sequences of KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures over a pool of five frames (some of them
    edited in place between calls), with and without sequential mode: a tracking context that remembers what its slots hold gives
    the same lists, call by call, as one that forgets before every call (= the reference's behaviour of converting and rebuilding
    everything every time)."""
    from helpers import make_tc
    sgf, trk = _api_modules()
    rng = np.random.default_rng(5)
    try:
        base = synth.synth_base(400, 300, 8)
        pool = [synth.synth_frame(400, 300, 8, k, shift=(1.3, 0.9), base=base) for k in range(5)]
        for trial in range(40):
            seq_mode = bool(trial & 1)
            frames_a = [f.copy() for f in pool]
            frames_b = [f.copy() for f in pool]
            tcs = []
            for _ in range(2):
                tc = make_tc(levels=2, ss=2, max_residue=12.0)
                tc.sequentialMode = seq_mode
                tcs.append(tc)
            n = int(rng.integers(40, 200))
            i0 = int(rng.integers(0, 5))
            fls = [sgf.KLTSelectGoodFeatures(tcs[0], frames_a[i0], n)]
            trk.KLTForgetFrames(tcs[1])
            fls.append(sgf.KLTSelectGoodFeatures(tcs[1], frames_b[i0], n))
            assert _records(fls[0]) == _records(fls[1])
            cur = i0
            for step in range(10):
                op = rng.choice(["track", "track", "track", "replace", "select", "edit"])
                if op == "edit":                               # same object, new pixels (a block large enough to hold lattice samples)
                    k = int(rng.integers(0, 5))
                    y, x = int(rng.integers(0, 200)), int(rng.integers(0, 300))
                    val = int(rng.integers(0, 255))
                    for fr in (frames_a, frames_b):
                        fr[k][y:y + 60, x:x + 60] = val
                    continue
                nxt = int(rng.integers(0, 5))
                for which, (tc, fr) in enumerate(zip(tcs, (frames_a, frames_b))):
                    if which == 1:
                        trk.KLTForgetFrames(tc)
                    if op == "track":
                        trk.KLTTrackFeatures(tc, fr[cur], fr[nxt], fls[which])
                    elif op == "replace":
                        sgf.KLTReplaceLostFeatures(tc, fr[cur], fls[which])
                    else:
                        fls[which] = sgf.KLTSelectGoodFeatures(tc, fr[nxt], n)
                if op != "replace":
                    cur = nxt
                assert _records(fls[0]) == _records(fls[1]), "trial %d step %d (%s, sequential %s)" % (trial, step, op, seq_mode)
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = 1


def test_klt_pyramid_class_on_the_device(cfg1, synth251):
    """KLTPyramid.Compute (pyramid.py:37-77; an API-compatibility class, the tracker builds its pyramids with klt_build_pyramids):
    all levels in one call on the device, equal to the reference's pyramid planes -- level 1 of img0's and img1's image pyramids from
    their level 0 (goldens of the reference itself), and to the levels of the tracker's own build at an odd size with three levels."""
    from helpers import make_tc, synth251_frames
    from pyfeaturetrack_amd.backend import Context
    from pyfeaturetrack_amd.pyramid import KLTPyramid
    for name in ("p0", "p1"):
        lvl0 = cfg1[name + "_img_0"]
        pyr = KLTPyramid(lvl0.shape[1], lvl0.shape[0], 4, 2)
        pyr.Compute(lvl0, 0.9)
        assert np.array_equal(pyr.img[0], lvl0) and np.array_equal(pyr.img[1], cfg1[name + "_img_1"])
        assert pyr.ncols == [320, 80.0] and pyr.nrows == [240, 60.0]
    tc = make_tc(levels=3, ss=2)
    c = Context(0)
    try:
        c.configure(tc)
        c.upload(0, synth251_frames()[0])
        c.build_pyramids(0)
        want = [c.download_level(0, 0, l) for l in range(3)]
    finally:
        c.close()
    pyr = KLTPyramid(251, 187, 2, 3)
    pyr.Compute(want[0], tc.pyramid_sigma_fact)
    for l in range(3):
        assert np.array_equal(pyr.img[l], want[l]), l
    one = KLTPyramid(64, 48, 4, 1)
    one.Compute(np.ones((48, 64), np.float32), 0.9)
    assert len(one.img) == 1


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
