"""The reference-shaped Python API as a caller sees it (cfg-2's size: 1920x1080 numpy frames, 5000 features): ms per KLTSelectGoodFeatures /
KLTTrackFeatures call in the scenarios of VERDICT r4 next-1, and KLTTrackSequence per frame over 256 frames at 1080p and 4K.  Runs in a
process of its own -- `python -m benchlib.api_figures` prints one JSON object; bench.py starts it as a child and merges the keys into
`extra` (benchlib/cfg2.py)."""
from .common import *            # noqa: F401,F403


def api_figures(pair, tc):
    """What a caller of the reference-shaped Python API sees (KLTSelectGoodFeatures / KLTTrackFeatures on PIL-like arrays, uploads
    and the download of the list included): ms per call at cfg-2's size, on the package's default context."""
    from pyfeaturetrack_amd import selectGoodFeatures as sgf
    from pyfeaturetrack_amd import trackFeatures as trk
    v0 = sgf.KLT_verbose
    sgf.KLT_verbose = trk.KLT_verbose = 0
    try:
        a0, a1 = pair

        def as_pil(arr):
            """a mode-"L" Pillow image with storage of its own, as Image.open(...) of a PGM / PNG file gives (Image.fromarray would
            share the array's memory) -- what a script written against the reference passes (trackFeatures.py:165,176)"""
            from PIL import Image
            return Image.frombytes("L", (arr.shape[1], arr.shape[0]), arr.tobytes())

        def as_rgb(arr):
            """a colour frame (what a camera or a video decoder hands a script: the reference's GUI example feeds such frames): three
            different functions of the grey frame, storage of its own"""
            from PIL import Image
            rgb = np.dstack([arr, np.roll(arr, 3, axis=1), 255 - arr // 2]).astype(np.uint8)
            return Image.frombytes("RGB", (arr.shape[1], arr.shape[0]), rgb.tobytes())

        def measure(trusting, new_frame_per_call=False, pil=False):
            tc.trustFrameIdentity = trusting
            trk.KLTForgetFrames(tc)
            t_sel, t_trk, t_pp = [], [], []
            wrap = as_rgb if pil == "rgb" else as_pil
            f0, f1 = (wrap(a0), wrap(a1)) if pil else (a0, a1)
            fl = sgf.KLTSelectGoodFeatures(tc, f0, NFEAT)
            trk.KLTTrackFeatures(tc, f0, f1, fl)
            g1 = wrap(a1) if pil else f1.copy()
            for k in range(30):
                t = time.perf_counter()
                fl = sgf.KLTSelectGoodFeatures(tc, f0, NFEAT)
                t_sel.append(time.perf_counter() - t)
                if new_frame_per_call:                         # one pixel: frame 2 is a new image every call
                    if pil == "rgb":
                        r, g, b = g1.getpixel((k, k))
                        g1.putpixel((k, k), (r ^ 1, g, b))
                    elif pil:
                        g1.putpixel((k, k), g1.getpixel((k, k)) ^ 1)
                    else:
                        g1[k, k] ^= 1
                t = time.perf_counter()
                trk.KLTTrackFeatures(tc, f0, g1 if new_frame_per_call else f1, fl)
                t_trk.append(time.perf_counter() - t)
            # example1's ping-pong (example1.py:53-56): the same two images, back and forth
            fl = sgf.KLTSelectGoodFeatures(tc, f0, NFEAT)
            for k in range(40):
                a, b = (f0, f1) if k % 2 == 0 else (f1, f0)
                t = time.perf_counter()
                trk.KLTTrackFeatures(tc, a, b, fl)
                t_pp.append(time.perf_counter() - t)
            return statistics.median(t_sel) * 1e3, statistics.median(t_trk) * 1e3, statistics.median(t_pp) * 1e3

        f0, f1 = a0, a1

        def clip_loop(pil=False):
            # consecutive frames of a clip in non-sequential mode: frame 1 of a call is frame 2 of the call before, frame 2 has new
            # pixels (16 distinct frames visited up and down) -- the call a video loop written against the reference makes
            base = synth.synth_base(f0.shape[1], f0.shape[0], 1)
            clip = [synth.synth_frame(f0.shape[1], f0.shape[0], 1, k, base=base) for k in range(16)]
            if pil:
                clip = [as_pil(c) for c in clip]
            order = list(range(16)) + list(range(14, 0, -1))
            tc.trustFrameIdentity = False
            trk.KLTForgetFrames(tc)
            fl = sgf.KLTSelectGoodFeatures(tc, clip[0], NFEAT)
            ts = []
            for k in range(64):
                a, b = clip[order[k % 30]], clip[order[(k + 1) % 30]]
                t = time.perf_counter()
                trk.KLTTrackFeatures(tc, a, b, fl)
                ts.append(time.perf_counter() - t)
                if k % 8 == 7:
                    fl = sgf.KLTSelectGoodFeatures(tc, b, NFEAT)
            return statistics.median(ts[4:]) * 1e3

        def sequence(w, h, n, seed, nframes=256):
            # KLTTrackSequence itself (the product's sequence function; VERDICT r4 missing-4): numpy frames in, feature table out
            from pyfeaturetrack_amd.klt import KLT_TrackingContext
            from pyfeaturetrack_amd.trackSequence import KLTTrackSequence
            tcs = KLT_TrackingContext()
            tcs.nPyramidLevels, tcs.subsampling = 3, 4
            tcs.KLTUpdateTCBorder()
            tcs.max_residue = 10.0
            base = synth.synth_base(w, h, seed)
            distinct = [synth.synth_frame(w, h, seed, k, base=base) for k in range(16)]
            order = list(range(16)) + list(range(14, 0, -1))
            frames = [distinct[order[k % 30]] for k in range(nframes)]
            best = None
            for _ in range(3):
                t = time.perf_counter()
                KLTTrackSequence(tcs, frames, n)
                ms = (time.perf_counter() - t) * 1e3 / (nframes - 1)
                best = ms if best is None else min(best, ms)
            return best

        def sequential_loop(nframes=64):
            # the loop a video script written against the reference runs: sequential mode, per frame KLTTrackFeatures + KLTReplaceLostFeatures
            from pyfeaturetrack_amd.klt import KLT_TrackingContext
            tcs = KLT_TrackingContext()
            tcs.nPyramidLevels, tcs.subsampling = 3, 4
            tcs.KLTUpdateTCBorder()
            tcs.sequentialMode = True
            tcs.max_residue = 10.0
            base = synth.synth_base(f0.shape[1], f0.shape[0], 1)
            clip = [synth.synth_frame(f0.shape[1], f0.shape[0], 1, k, base=base) for k in range(16)]
            order = list(range(16)) + list(range(14, 0, -1))
            fl = sgf.KLTSelectGoodFeatures(tcs, clip[0], NFEAT)
            best = None
            for rep in range(6):                                   # (the first pass is the warm-up: slots, pinned buffers, score sets)
                t = time.perf_counter()
                for k in range(1, nframes):
                    prev, cur = clip[order[(k - 1) % 30]], clip[order[k % 30]]
                    trk.KLTTrackFeatures(tcs, prev, cur, fl)
                    sgf.KLTReplaceLostFeatures(tcs, cur, fl)
                ms = (time.perf_counter() - t) * 1e3 / (nframes - 1)
                best = ms if best is None or rep == 1 else min(best, ms)
            return best

        exact, trusting, fresh = measure(False), measure(True), measure(False, True)
        pil_exact, pil_fresh = measure(False, pil=True), measure(False, True, pil=True)
        tc.trustFrameIdentity = False
        from pyfeaturetrack_amd import _pil
        # ... and what the same calls cost when every image is first made into an array (np.asarray(img), the path of rounds 1-5 and the
        # fallback when the self-check of the row tables fails)
        rgb_exact, rgb_fresh = measure(False, pil="rgb"), measure(False, True, pil="rgb")
        was, _pil._layout = _pil.layout(), False
        try:
            conv_exact, conv_fresh = measure(False, pil=True), measure(False, True, pil=True)
            tc.trustFrameIdentity = False
            trk.KLTForgetFrames(tc)
            c0, c1 = as_rgb(a0), as_rgb(a1)
            fl_c = sgf.KLTSelectGoodFeatures(tc, c0, NFEAT)
            t_c = []
            for k in range(6):                                 # (a few calls are enough: Pillow's own conversion takes milliseconds)
                x, y = (c0, c1) if k % 2 == 0 else (c1, c0)
                t = time.perf_counter()
                trk.KLTTrackFeatures(tc, x, y, fl_c)
                t_c.append(time.perf_counter() - t)
            conv_rgb_pp = statistics.median(t_c) * 1e3
        finally:
            _pil._layout = was
        return {"api_ms_per_KLTSelectGoodFeatures_pil_rgb": rgb_exact[0], "api_ms_per_KLTTrackFeatures_pingpong_pil_rgb": rgb_exact[2],
                "api_ms_per_KLTTrackFeatures_new_frame_each_call_pil_rgb": rgb_fresh[1],
                "api_pil_rgb_note": "the same on COLOUR Pillow images (\"RGB\", 1920x1080): the reference converts them with img.convert(\"F\") -- Pillow's luma -- on every "
                                    "call; here the 4-byte pixels are compared as stored and the float frame is made from the image's rows by klt_host_luma_rows "
                                    "and sent with klt_upload_f32_async (8 MB per frame on the link instead of 2); with np.array(img.convert(\"F\")) per image and "
                                    "call (rounds 1-5) the ping-pong call reads %.2f ms" % conv_rgb_pp,
                "api_pil_converted_to_arrays": {"api_ms_per_KLTSelectGoodFeatures_pil": conv_exact[0], "api_ms_per_KLTTrackFeatures_pil": conv_exact[1],
                                                "api_ms_per_KLTTrackFeatures_pingpong_pil": conv_exact[2],
                                                "api_ms_per_KLTTrackFeatures_new_frame_each_call_pil": conv_fresh[1],
                                                "note": "row tables switched off (as KLT_NO_PIL_ROWS=1): np.asarray(img) per image and call"},
                "api_ms_per_KLTSelectGoodFeatures_pil": pil_exact[0], "api_ms_per_KLTTrackFeatures_pil": pil_exact[1],
                "api_ms_per_KLTTrackFeatures_pingpong_pil": pil_exact[2],
                "api_ms_per_KLTTrackFeatures_new_frame_each_call_pil": pil_fresh[1],
                "api_ms_per_KLTTrackFeatures_consecutive_frames_pil": clip_loop(pil=True),
                "api_pil_note": "the same scenarios with mode-\"L\" Pillow images that own their storage (the reference's image type: "
                                "trackFeatures.py:165,176 `img.convert(\"F\")`); the images are read through Pillow's row table (Image.getim(), "
                                "pyfeaturetrack_amd/_pil.py: %s) -- no array is made of them; new_frame_each_call edits frame 2 with putpixel"
                                % ("active, struct layout found by the start-up self-check" if _pil.status()["active"] else
                                   "INACTIVE (%s): np.asarray(img) per image and call" % _pil.status()["why_not"]),
                "api_ms_per_KLTSelectGoodFeatures": exact[0], "api_ms_per_KLTTrackFeatures": exact[1],
                "api_ms_per_KLTTrackFeatures_pingpong": exact[2],
                "api_ms_per_KLTTrackFeatures_new_frame_each_call": fresh[1],
                "api_ms_per_KLTTrackFeatures_consecutive_frames": clip_loop(),
                "api_ms_per_frame_sequential_mode_loop": sequential_loop(),
                "api_ms_per_frame_KLTTrackSequence": {"1080p_5000_features_256_frames": sequence(1920, 1080, 5000, 1),
                                                      "4k_20000_features_256_frames": sequence(3840, 2160, 20000, 4),
                                                      "note": "the whole call (first selection, helper thread, table download) / 255; "
                                                              "replacement after every frame; best of 3"},
                "api_trusting_ms_per_KLTSelectGoodFeatures": trusting[0], "api_trusting_ms_per_KLTTrackFeatures": trusting[1],
                "api_trusting_ms_per_KLTTrackFeatures_pingpong": trusting[2],
                "api_note": "reference-shaped Python API on numpy u8 frames of cfg-2's size, 5000 features; host-to-device copies and the "
                            "download of the list are inside the figures.  api_* = the default: a frame is reused only after EVERY byte "
                            "was compared with the copy the slot was filled from (results identical to the reference's for any call "
                            "sequence); api_trusting_* = the opt-in tc.trustFrameIdentity shortcut (object identity + 1024 sampled pixels); "
                            "new_frame_each_call = frame 2 differs by one pixel in every call (compare, copy to pinned memory, DMA, pyramid, track); "
                            "consecutive_frames = a clip walked pair by pair in non-sequential mode (frame 1 resident from the call before, frame 2 new); "
                            "sequential_mode_loop = tc.sequentialMode, per frame KLTTrackFeatures + KLTReplaceLostFeatures (63 frames, best of 5 after a warm-up pass); "
                            "per-call figures are medians of 30-64 calls"}
    finally:
        sgf.KLT_verbose = trk.KLT_verbose = v0



if __name__ == "__main__":
    from .common import cfg2_context
    if os.environ.get("KLT_API_FIGURES_CPUS"):
        os.sched_setaffinity(0, {int(c) for c in os.environ["KLT_API_FIGURES_CPUS"].split(",")})
    _fd = os.dup(1)                      # the HIP runtime / RCCL print to fd 1: the JSON goes to the saved descriptor
    os.dup2(2, 1)
    _tc = cfg2_context()
    _pair = synth.synth_pair(WIDTH, HEIGHT, seed=1)
    _out = api_figures(_pair, _tc)
    os.write(_fd, (json.dumps(_out) + "\n").encode())
