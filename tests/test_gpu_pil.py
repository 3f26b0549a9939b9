"""The reference's own image type on the API path: KLTSelectGoodFeatures / KLTTrackFeatures / KLTReplaceLostFeatures /
KLTTrackSequence on mode-"L" Pillow images (selectGoodFeatures.py:190, trackFeatures.py:165,176: `img.convert("F")`) give what the same
calls give on arrays of the same pixels -- with the images read through Pillow's row table (pyfeaturetrack_amd/_pil.py) and with that
switched off --, an image edited in place with putpixel between two calls is a new frame, and none of it makes an array of an image.
Colour ("RGB" / "RGBA") and float ("F") images: against what the reference gave on them (tests/golden/colour_images.npz), without Pillow's
conversion on the way."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import make_tc
from pyfeaturetrack_amd import synth

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")
# tests that assert the row-table path itself are meaningless (not wrong) when the environment switches it off
rows_path = pytest.mark.skipif(os.environ.get("KLT_NO_PIL_ROWS") == "1", reason="Pillow's row tables were switched off through the environment")


def _api():
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import trackFeatures as trk
    sgf.KLT_verbose = trk.KLT_verbose = 0
    return sgf, trk


def _records(fl):
    return [(f.x, f.y, f.val) for f in fl]


def _owned(arr):
    """a Pillow image with storage of its own (what Image.open gives), not a view of the array"""
    return Image.frombytes("L", (arr.shape[1], arr.shape[0]), arr.tobytes())


@pytest.fixture()
def no_array_of_a_pil_image(monkeypatch):
    """fails the test if an 8-bit Pillow image is converted to an array anywhere on the way"""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    real = sgf.image_to_array

    def guarded(img):
        assert isinstance(img, np.ndarray) or getattr(img, "mode", None) != "L", "np.asarray(img) of an 8-bit Pillow image"
        return real(img)
    monkeypatch.setattr(sgf, "image_to_array", guarded)


@rows_path
def test_row_tables_are_active_on_the_gpu_box():
    from pyfeaturetrack_amd import _pil
    st = _pil.status()
    assert st["active"], st["why_not"]


@rows_path
def test_example1_on_pillow_images_gives_the_reference_lists(cfg1, no_array_of_a_pil_image):
    """example1.py:40-56 as the reference runs it -- Image.open of the two PGM files -- : selection and tracked list of cfg-1"""
    sgf, trk = _api()
    img0, img1 = Image.open(os.path.join(GOLDEN, "img0.pgm")), Image.open(os.path.join(GOLDEN, "img1.pgm"))
    assert img0.mode == "L"
    tc = make_tc(max_residue=10.0)
    fl = sgf.KLTSelectGoodFeatures(tc, img0, 100)
    assert [f.x for f in fl] == cfg1["sel100_x"].tolist() and [f.y for f in fl] == cfg1["sel100_y"].tolist() and [f.val for f in fl] == cfg1["sel100_val"].tolist()
    trk.KLTTrackFeatures(tc, img0, img1, fl)
    assert np.array_equal(np.array([f.val for f in fl]), cfg1["trk100_r10_val"])
    assert np.array_equal(np.array([f.x for f in fl], np.float64), cfg1["trk100_r10_x"]) and np.array_equal(np.array([f.y for f in fl], np.float64), cfg1["trk100_r10_y"])


@pytest.mark.parametrize("rows_on", [True, False])
def test_pillow_images_give_what_arrays_give(rows_on, monkeypatch):
    """the three public calls over a clip, once on numpy frames and once on Pillow images (owned storage and array-mapped ones
    alternating), sequential mode included; `rows_on` False = the conversion path (np.asarray per image)"""
    from pyfeaturetrack_amd import _pil
    if not rows_on:
        monkeypatch.setattr(_pil, "_layout", False)
    sgf, trk = _api()
    w, h, n = 648, 486, 300
    base = synth.synth_base(w, h, 41)
    frames = [synth.synth_frame(w, h, 41, r, shift=(1.7, -1.1), base=base) for r in range(7)]
    pils = [_owned(f) if k % 2 == 0 else Image.fromarray(f) for k, f in enumerate(frames)]
    assert (_pil.rows_of(pils[0]) is not None) == (rows_on and os.environ.get("KLT_NO_PIL_ROWS") != "1")

    def run(imgs, sequential):
        tc = make_tc(levels=3, ss=2, window=9, max_residue=10.0)
        tc.sequentialMode = sequential
        out = []
        fl = sgf.KLTSelectGoodFeatures(tc, imgs[0], n)
        out.append(_records(fl))
        for k in range(1, len(imgs)):
            trk.KLTTrackFeatures(tc, imgs[k - 1], imgs[k], fl)
            out.append(_records(fl))
            sgf.KLTReplaceLostFeatures(tc, imgs[k], fl)
            out.append(_records(fl))
        trk.KLTTrackFeatures(tc, imgs[-1], imgs[0], fl)                      # ... and back to a frame sent long ago
        out.append(_records(fl))
        return out

    for sequential in (False, True):
        assert run(pils, sequential) == run(frames, sequential), "sequential mode %s" % sequential


@rows_path
def test_an_image_edited_with_putpixel_between_calls_is_a_new_frame(no_array_of_a_pil_image):
    """`putpixel` on the very object a slot was filled from -- on the 1024-pixel lattice, off it, and a whole block -- then the call again:
    the results are those of fresh images with these pixels (the reference converts the image anew on every call)"""
    sgf, trk = _api()
    w, h, n = 648, 486, 300
    base = synth.synth_base(w, h, 43)
    f0, f1 = (synth.synth_frame(w, h, 43, r, shift=(1.7, -1.1), base=base) for r in (0, 1))
    tc = make_tc(levels=3, ss=2, window=9)
    img0, img1 = _owned(f0), _owned(f1)

    def fresh(a0, a1):
        tcf = make_tc(levels=3, ss=2, window=9)
        fl = sgf.KLTSelectGoodFeatures(tcf, a0, n)
        trk.KLTTrackFeatures(tcf, a0, a1, fl)
        return _records(fl)

    fl = sgf.KLTSelectGoodFeatures(tc, img0, n)
    trk.KLTTrackFeatures(tc, img0, img1, fl)
    assert _records(fl) == fresh(f0, f1)
    edits = {"a lattice pixel": [(0, 0)], "one pixel off the lattice": [(301, 211)],
             "a block between lattice samples": [(x, y) for y in range(200, 209) for x in range(321, 330)]}
    for which, img, arr in (("frame 2", img1, f1), ("frame 1", img0, f0)):
        for name, pts in edits.items():
            for (x, y) in pts:
                v = (int(arr[y, x]) + 90) % 256
                img.putpixel((x, y), v)
                arr[y, x] = v
            fl = sgf.KLTSelectGoodFeatures(tc, img0, n)
            trk.KLTTrackFeatures(tc, img0, img1, fl)
            assert _records(fl) == fresh(f0.copy(), f1.copy()), "%s after editing %s" % (which, name)


@rows_path
def test_track_sequence_on_pillow_images(no_array_of_a_pil_image):
    """KLTTrackSequence over Pillow images: the helper thread stages them from their row tables; the table equals the one numpy frames give"""
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    w, h, n = 648, 486, 250
    base = synth.synth_base(w, h, 47)
    frames = [synth.synth_frame(w, h, 47, r, shift=(1.7, -1.1), base=base) for r in range(12)]
    want = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), frames, n)
    got = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), [_owned(f) for f in frames], n)
    assert np.array_equal(want.rec, got.rec)


def test_float_grey_colour_and_other_pillow_modes_agree(cfg1):
    """a mode-"F" copy of img0 and a grey "RGB" image give img0's lists (float and colour rows: section below); modes without a row-table
    path ("I", "P") are converted with img.convert("F") as the reference does"""
    sgf, trk = _api()
    img0 = Image.open(os.path.join(GOLDEN, "img0.pgm"))
    tc = make_tc(max_residue=10.0)
    fl = sgf.KLTSelectGoodFeatures(tc, img0.convert("F"), 100)
    assert [f.x for f in fl] == cfg1["sel100_x"].tolist() and [f.val for f in fl] == cfg1["sel100_val"].tolist()
    grey_rgb = Image.merge("RGB", (img0, img0, img0))
    want = np.array(grey_rgb.convert("F"))
    fl2 = sgf.KLTSelectGoodFeatures(make_tc(max_residue=10.0), grey_rgb, 100)
    fl3 = sgf.KLTSelectGoodFeatures(make_tc(max_residue=10.0), want, 100)
    assert _records(fl2) == _records(fl3)
    for other in (img0.convert("I"), img0.convert("P")):
        fl4 = sgf.KLTSelectGoodFeatures(make_tc(max_residue=10.0), other, 100)
        fl5 = sgf.KLTSelectGoodFeatures(make_tc(max_residue=10.0), np.array(other.convert("F")), 100)
        assert _records(fl4) == _records(fl5), other.mode


# ------------------------------------------------------------------------------------------------ colour and float images
def _colour_cases(img0, img1):
    from helpers import colour_of
    c0, c1 = colour_of(img0), colour_of(img1)
    alpha = (np.arange(img0.size, dtype=np.uint32).reshape(img0.shape) * 7 % 256).astype(np.uint8)
    return {"rgb": (Image.fromarray(c0, "RGB"), Image.fromarray(c1, "RGB")),
            "rgba": (Image.fromarray(np.dstack([c0, alpha]), "RGBA"), Image.fromarray(np.dstack([c1, alpha]), "RGBA")),
            "f": (Image.fromarray(c0, "RGB").convert("F"), Image.fromarray(c1, "RGB").convert("F"))}


@rows_path
@pytest.mark.parametrize("mode", ["rgb", "rgba", "f"])
def test_colour_and_float_images_give_the_reference_lists(mode, img0, img1, golden_dir):
    """KLTSelectGoodFeatures / KLTTrackFeatures on "RGB", "RGBA" and "F" Pillow images against what the REFERENCE gave on the same images
    (tests/golden/gen_colour_images.py): selection, track, track back -- with Pillow's own conversion out of the way (the float frame is
    made from the image's rows by klt_host_luma_rows and sent with klt_upload_f32_async)."""
    g = np.load(os.path.join(golden_dir, "colour_images.npz"))
    sgf, trk = _api()
    i0, i1 = _colour_cases(img0, img1)[mode]
    tc = make_tc(max_residue=10.0)

    def rec(fl):
        return np.array([(f.x, f.y, f.val) for f in fl], np.float64)

    undo = pytest.MonkeyPatch()
    undo.setattr(Image.Image, "convert", lambda self, *a, **k: (_ for _ in ()).throw(AssertionError("img.convert%r of a %s image" % (a, self.mode))))
    try:
        fl = sgf.KLTSelectGoodFeatures(tc, i0, 100)
        assert np.array_equal(rec(fl), g[mode + "_sel"])
        trk.KLTTrackFeatures(tc, i0, i1, fl)
        assert np.array_equal(rec(fl), g[mode + "_trk"])
        trk.KLTTrackFeatures(tc, i1, i0, fl)                                   # both frames resident: compared as stored, nothing converted or sent
        assert np.array_equal(rec(fl), g[mode + "_back"])
    finally:
        undo.undo()


def test_colour_images_over_a_clip_equal_their_float_frames(monkeypatch):
    """a colour clip through the three public calls (sequential mode too), a frame edited with putpixel between two calls, KLTTrackSequence
    on colour frames: everything equals the same calls on `np.array(img.convert("F"))` of the same images -- the path of rounds 1-5, which
    the switch KLT_NO_PIL_ROWS still selects"""
    from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
    sgf, trk = _api()
    w, h, n = 648, 486, 300
    base = synth.synth_base(w, h, 51)
    grey = [synth.synth_frame(w, h, 51, r, shift=(1.7, -1.1), base=base) for r in range(6)]
    colour = [Image.fromarray(np.dstack([g, np.roll(g, 2, axis=0), 255 - g]), "RGB") for g in grey]
    floats = [np.array(c.convert("F")) for c in colour]

    def run(imgs, sequential, edit=None):
        tc = make_tc(levels=3, ss=2, window=9, max_residue=10.0)
        tc.sequentialMode = sequential
        out = []
        fl = sgf.KLTSelectGoodFeatures(tc, imgs[0], n)
        out.append(_records(fl))
        for k in range(1, len(imgs)):
            if edit and k == 3:
                edit(imgs[k])
            trk.KLTTrackFeatures(tc, imgs[k - 1], imgs[k], fl)
            out.append(_records(fl))
            sgf.KLTReplaceLostFeatures(tc, imgs[k], fl)
            out.append(_records(fl))
        return out

    for sequential in (False, True):
        assert run(colour, sequential) == run(floats, sequential), "sequential mode %s" % sequential
    # an edit in place: the image object is the one frame 3 was ... the call before compared; its float frame is made anew
    edited = [c.copy() for c in colour]
    want = [f.copy() for f in floats]
    patch = Image.new("RGB", (40, 30), (250, 10, 90))

    def edit_pil(im):
        im.paste(patch, (300, 200))
    ref_img = colour[3].copy()
    ref_img.paste(patch, (300, 200))
    want[3] = np.array(ref_img.convert("F"))
    got = run(edited, False, edit=edit_pil)
    assert got == run(want, False)
    a = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), colour, n)       # float staging buffers, filled by the helper thread
    b = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), floats, n)
    c = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), colour, n, async_ingest=False)
    mixed = [colour[0], floats[1], colour[2].convert("F"), colour[3], floats[4], colour[5]]
    d = KLTTrackSequence(make_tc(levels=3, ss=2, window=9, max_residue=10.0), mixed, n)
    assert np.array_equal(a.rec, b.rec) and np.array_equal(a.rec, c.rec) and np.array_equal(a.rec, d.rec)
