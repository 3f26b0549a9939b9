"""Affine consistency check (BASELINE cfg-3) -- PARITY UNPINNED.

The reference calls the affine tracker but does not define it, so there are no golden vectors.  These tests pin
behaviour with synthetic known answers (a texture warped by a known similarity / affine map) on the oracle (CPU
tests) and on the HIP path (GPU tests), and compare the two implementations with each other.
"""
import math

import numpy as np
import pytest

from helpers import make_tc, params_from_tc
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.params import affine_params_from_tc

W, H, NF = 640, 480, 300


def rot_scale(deg, scale):
    c, s = math.cos(math.radians(deg)) * scale, math.sin(math.radians(deg)) * scale
    return np.array([[c, -s], [s, c]])


def warped_sequence(A_step, t_step, nframes, seed=9):
    base = synth.synth_base(W, H, seed, sigma=2.5)
    frames, A = [], np.eye(2)
    for k in range(nframes):
        frames.append(synth.warp_frame(base, A, (k * t_step[0], k * t_step[1])))
        A = A_step @ A
    return frames


def tc_affine(mode, **kw):
    tc = make_tc(levels=2, ss=2, max_residue=12.0)
    tc.affineConsistencyCheck = mode
    tc.affine_max_residue = 12.0
    for k, v in kw.items():
        setattr(tc, k, v)
    return tc


def run_oracle(frames, tc, nf=NF):
    from oracle import klt_oracle as ko
    p, ap = params_from_tc(tc), affine_params_from_tc(tc)
    f32 = [f.astype(np.float32) for f in frames]
    fl = ko.select_good_features(p, f32[0], nf)
    st = ko.AffineState(ap, nf)
    P = [ko.Pyramids(p, f) for f in f32]
    hist = []
    for k in range(1, len(frames)):
        ko.track_features_affine(p, P[k - 1], P[k], fl, st)
        hist.append((fl.copy(), st.rec.copy()))
    return hist


def test_oracle_pure_translation_keeps_identity():
    frames = warped_sequence(np.eye(2), (1.3, -0.8), 4)
    hist = run_oracle(frames, tc_affine(2))
    fl1, rec1 = hist[0]
    assert np.all(rec1["valid"][fl1["val"] == 0] == 1) and np.all(rec1["Axx"] == 1)       # first call only stores templates
    fl3, rec3 = hist[-1]
    live = fl3["val"] == 0
    assert live.sum() > 0.9 * NF
    for name, want in (("Axx", 1), ("Ayy", 1), ("Axy", 0), ("Ayx", 0)):
        assert np.abs(rec3[name][live] - want).max() < 0.03, name
    assert np.all(rec3["valid"][~live & (fl3["val"] < 0)] == 0)


@pytest.mark.parametrize("mode", [1, 2])
def test_oracle_recovers_known_similarity(mode):
    A_step = rot_scale(0.6, 1.004)
    frames = warped_sequence(A_step, (0.7, 0.4), 4)
    hist = run_oracle(frames, tc_affine(mode))
    fl, rec = hist[-1]
    live = fl["val"] == 0
    assert live.sum() > 0.8 * NF
    want = np.linalg.matrix_power(A_step, 3)          # template cut from frame 0, tracked into frame 3
    got = np.array([[np.median(rec["Axx"][live]), np.median(rec["Axy"][live])],
                    [np.median(rec["Ayx"][live]), np.median(rec["Ayy"][live])]])
    assert np.abs(got - want).max() < 4e-3, (got, want)


def test_oracle_rejects_inconsistent_feature():
    """a region that changes appearance (here: replaced by different texture) fails the consistency check"""
    frames = warped_sequence(np.eye(2), (1.0, 0.5), 3)
    other = synth.synth_base(W, H, 77, sigma=2.5)
    bad = frames[2].copy()
    bad[100:380, 100:540] = np.clip(0.5 * bad[100:380, 100:540] + 0.5 * other[100:380, 100:540], 0, 255).astype(np.uint8)
    hist = run_oracle([frames[0], frames[1], bad], tc_affine(2, affine_max_residue=6.0, max_residue=None))
    fl, rec = hist[-1]
    x0, y0 = hist[0][0]["x"], hist[0][0]["y"]
    inside = (x0 > 130) & (x0 < 510) & (y0 > 130) & (y0 < 350) & (hist[0][0]["val"] == 0)
    outside = ((x0 < 80) | (x0 > 560) | (y0 < 80) | (y0 > 400)) & (hist[0][0]["val"] == 0)
    assert np.mean(fl["val"][inside] < 0) > 0.8
    assert np.mean(fl["val"][outside] == 0) > 0.9


# ------------------------------------------------------------------------------------------- GPU
def run_gpu(frames, tc, nf=NF):
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        c.configure(tc)
        for k, f in enumerate(frames):
            c.upload(k, f)
        c.build_pyramids_batch(list(range(len(frames))), sync=True)
        fl, _ = c.select(0, nf, use_pyramid=True)
        c.affine_alloc(0, nf)
        hist = []
        for k in range(1, len(frames)):
            fl, _ = c.track_affine(k - 1, k, fl, 0)
            hist.append((fl.copy(), c.affine_download(0, nf)))
        return hist
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gpu_affine_matches_oracle(mode):
    A_step = rot_scale(0.5, 1.003) if mode else np.eye(2)
    frames = warped_sequence(A_step, (0.9, -0.6), 4)
    tc = tc_affine(mode)
    g, o = run_gpu(frames, tc), run_oracle(frames, tc)
    # Exact: the affine check's window sums follow one stated order (per-lane partial sums + butterfly, DESIGN.md section 8;
    # oracle/klt_oracle.c am_fold), so every status code, position and A matrix is bit-identical -- integer work is not
    # allowed to be "mostly right".
    for k, ((gfl, grec), (ofl, orec)) in enumerate(zip(g, o)):
        assert np.array_equal(gfl["val"], ofl["val"]), "call %d: %d status codes differ" % (k, int((gfl["val"] != ofl["val"]).sum()))
        assert np.array_equal(gfl["x"], ofl["x"]) and np.array_equal(gfl["y"], ofl["y"]), "call %d: positions" % k
        for name in ("valid", "aff_x", "aff_y", "Axx", "Ayx", "Axy", "Ayy"):
            live = orec["valid"] != 0                    # the A entries of a freed template are don't-cares upstream too; keep them equal anyway
            assert np.array_equal(grec[name][live], orec[name][live]), (k, name)
            assert np.array_equal(grec[name], orec[name]), (k, name, "freed templates")
    assert (g[-1][0]["val"] == 0).sum() > 0.5 * NF


@pytest.mark.gpu
def test_gpu_affine_recovers_known_affine_map():
    A_step = np.array([[1.004, 0.006], [-0.003, 0.997]])
    frames = warped_sequence(A_step, (0.5, 0.8), 4)
    hist = run_gpu(frames, tc_affine(2))
    fl, rec = hist[-1]
    live = fl["val"] == 0
    assert live.sum() > 0.8 * NF
    want = np.linalg.matrix_power(A_step, 3)
    got = np.array([[np.median(rec["Axx"][live]), np.median(rec["Axy"][live])],
                    [np.median(rec["Ayx"][live]), np.median(rec["Ayy"][live])]])
    assert np.abs(got - want).max() < 4e-3, (got, want)


@pytest.mark.gpu
def test_gpu_affine_python_api_and_replacement(capsys):
    """KLTTrackFeatures with tc.affineConsistencyCheck = 2 through the reference-shaped API; replaced features start
    a new template (selectGoodFeatures.py:120-128)."""
    PIL = pytest.importorskip("PIL.Image")
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd.trackFeatures import KLTTrackFeatures
    frames = warped_sequence(rot_scale(0.4, 1.002), (1.1, 0.3), 4)
    imgs = [PIL.fromarray(f, "L") for f in frames]
    tc = tc_affine(2)
    tc.sequentialMode = True
    sgf.KLT_verbose = 0
    try:
        fl = sgf.KLTSelectGoodFeatures(tc, imgs[0], 120)
        KLTTrackFeatures(tc, imgs[0], imgs[1], fl)
        assert all(f.aff_img is not None and f.aff_Axx == 1.0 for f in fl if f.val == 0)
        fl[3].val = -4                                     # pretend a feature was lost
        fl[3].x = fl[3].y = -1.0
        sgf.KLTReplaceLostFeatures(tc, imgs[1], fl)
        assert fl[3].val > 0 and fl[3].aff_img is None
        KLTTrackFeatures(tc, imgs[1], imgs[2], fl)
        KLTTrackFeatures(tc, imgs[2], imgs[3], fl)
        live = [f for f in fl if f.val == 0]
        assert len(live) > 90
        assert abs(np.median([f.aff_Axx for f in live]) - 1.0) < 0.02
        assert all(f.aff_img is None for f in fl if f.val < 0)          # templates of lost features are freed (:292-341, :393-395)
    finally:
        sgf.KLT_verbose = 1


@pytest.mark.gpu
def test_gpu_affine_matches_oracle_on_random_draws():
    """tests/fuzz/fuzz_parity.py --affine: frame sizes, pyramid shapes, tracker windows, affine windows 9-21, modes 0 / 1 / 2, residue and
    displacement limits, iteration counts and a random affine map per frame -- 20 draws (380 were run when the tool was written): every
    status, position, template offset and matrix entry equals the oracle's after each of three calls."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from pyfeaturetrack_amd.backend import Context
    rng = np.random.default_rng(31)
    c = Context(0)
    try:
        alive = 0
        for k in range(20):
            t = fz.draw(rng, 300000, 400, 800)
            bad = fz.run_affine_trial(c, t)
            assert bad is None, "draw %d: %s differs from the oracle: %r" % (k, bad, t)
            alive += int(t["_stat"].split(", ")[2].split(" of ")[0])
        assert alive > 500
    finally:
        c.close()
