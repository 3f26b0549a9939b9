"""Out of memory is an answer, not a device error: every allocation of the library returns KLT_ERR_NOMEM with the size
asked for, leaves nothing half-allocated and no stale error with the HIP runtime, and the same context carries on -- with an injected
failure at every allocation site of a call sequence in turn (KLT_OPT_FAIL_ALLOC_AFTER) and with the device's memory really used up."""
import os

import numpy as np
import pytest

from helpers import baseline_case, make_tc

pytestmark = pytest.mark.gpu
KLT_OPT_FAIL_ALLOC_AFTER = 19
GB = 1 << 30


@pytest.fixture(scope="module")
def big(golden_dir):
    return np.load(os.path.join(golden_dir, "baseline_sizes.npz"))


def _golden(fl, g, tag, what):
    assert np.array_equal(fl["val"], g["%s_%s_val" % (tag, what)]), "%s %s: status / value words" % (tag, what)
    assert np.array_equal(fl["x"], g["%s_%s_x" % (tag, what)]) and np.array_equal(fl["y"], g["%s_%s_y" % (tag, what)]), "%s %s: positions" % (tag, what)


def test_an_injected_failure_at_every_allocation_site_leaves_the_context_usable(cfg1, img0, img1):
    """One context, the calls of a tracking script (uploads, pyramids, selection from the raw frame and from the pyramid, feature
    buffers, translation tracker, affine state + affine tracker, stand-alone convolutions, plane download, pinned memory): the k-th
    allocation from the start of the sequence is refused, k = 0, 1, 2, ... until a run gets through.  Every refusal is KLT_ERR_NOMEM
    naming a size; the call that was refused is simply made again (nothing else is repeated) and the results are the reference's."""
    from pyfeaturetrack_amd.backend import Context, KltOutOfMemory
    from pyfeaturetrack_amd.convolve import _computeKernels
    from pyfeaturetrack_amd.params import affine_params_from_tc
    tc = make_tc(max_residue=10.0)
    tca = make_tc(max_residue=10.0, affineConsistencyCheck=2)
    g1, d1 = _computeKernels(1.0)
    f0 = img0.astype(np.float32)

    def sequence(ctx, attempt):
        """`attempt(fn)` makes a call of the sequence; returns what the checks below look at"""
        res = {}
        attempt(lambda: ctx.configure(tc))
        attempt(lambda: ctx.upload(0, img0))
        attempt(lambda: ctx.upload(1, img1))
        attempt(lambda: ctx.build_pyramids_batch([0, 1], sync=True))
        res["sel_raw"] = attempt(lambda: ctx.select(0, 100, use_pyramid=False))[0]
        res["sel_pyr"] = attempt(lambda: ctx.select(0, 100, use_pyramid=True))[0]
        attempt(lambda: ctx.featbuf_upload(0, res["sel_pyr"]))
        attempt(lambda: ctx.track_async(0, 1, 0, 1, 100))
        res["trk"] = attempt(lambda: ctx.featbuf_download(1, 100))
        res["gx_plane"] = attempt(lambda: ctx.download_level(0, 1, 0))                 # (an interleaved plane leaves through a scratch plane)
        res["smooth"] = attempt(lambda: ctx.convolve_separate(f0, g1, g1))
        res["grads"] = attempt(lambda: ctx.gradients(f0, g1, d1))
        pin = attempt(lambda: ctx.pinned_array((240, 320)))
        pin[:] = img1
        attempt(lambda: ctx.upload_async(2, pin))
        attempt(lambda: ctx.build_pyramids(2, sync=True))
        attempt(lambda: ctx.configure(tca))
        attempt(lambda: ctx.build_pyramids_batch([0, 2], sync=True))
        attempt(lambda: ctx.affine_alloc(0, 100))
        attempt(lambda: ctx.featbuf_upload(5, res["sel_pyr"]))
        attempt(lambda: ctx.track_affine_async(0, 2, 5, 6, 100, 0))
        res["trk_affine"] = attempt(lambda: ctx.featbuf_download(6, 100))
        store = attempt(lambda: ctx.device_alloc(320 * 240))
        ctx.device_write(store, img0)
        attempt(lambda: ctx.adopt_u8(3, store, 320, 240))
        attempt(lambda: ctx.build_pyramids(3, sync=True))
        res["sel_adopted"] = attempt(lambda: ctx.select(3, 100, use_pyramid=True))[0]
        return res

    def check(res):
        for key in ("sel_raw", "sel_pyr", "sel_adopted"):
            assert np.array_equal(res[key]["x"], cfg1["sel100_x"]) and np.array_equal(res[key]["y"], cfg1["sel100_y"]) \
                and np.array_equal(res[key]["val"], cfg1["sel100_val"]), key
        for key in ("trk", "trk_affine"):      # (first affine call: templates stored, the translation result stands)
            assert np.array_equal(res[key]["val"], cfg1["trk100_r10_val"]) and np.array_equal(res[key]["x"].astype(np.float64), cfg1["trk100_r10_x"]) \
                and np.array_equal(res[key]["y"].astype(np.float64), cfg1["trk100_r10_y"]), key
        assert np.array_equal(res["gx_plane"], cfg1["p0_gx_0"])

    plain = Context(0)
    try:
        want = sequence(plain, lambda fn: fn())
        check(want)
    finally:
        plain.close()

    refused, sizes = [], []
    for k in range(200):
        ctx = Context(0)
        fired = []

        def attempt(fn, ctx=ctx, fired=fired):
            try:
                return fn()
            except KltOutOfMemory as e:
                assert "bytes asked for" in str(e) and "error -4" in str(e), str(e)
                assert isinstance(e, MemoryError)
                fired.append(str(e))
                return fn()                                        # the hook has fired (it disarms itself): the same call again

        try:
            ctx.set_option(KLT_OPT_FAIL_ALLOC_AFTER, k)
            got = sequence(ctx, attempt)
            ctx.sync()
            check(got)
            for key in ("smooth", "trk"):
                assert np.array_equal(got[key], want[key]), key
            assert np.array_equal(got["grads"][0], want["grads"][0]) and np.array_equal(got["grads"][1], want["grads"][1])
        finally:
            ctx.set_option(KLT_OPT_FAIL_ALLOC_AFTER, -1)
            ctx.close()
        if not fired:
            break
        assert len(fired) == 1
        refused.append(fired[0])
    else:
        pytest.fail("the sequence never ran out of allocation sites")
    assert k >= 20, "only %d allocations in the whole sequence?" % k
    what = {r.split("(")[-1].rstrip(")") for r in refused}
    assert len(what) >= 12, "allocation sites seen: %r" % sorted(what)


def test_device_memory_used_up_then_freed(big):
    """The device's memory really runs out (klt_device_alloc until it refuses, down to 8 MB pieces): klt_build_pyramids of a 4K frame answers
    KLT_ERR_NOMEM; after the pieces are freed the very next build + selection + track on the same context succeed and are the
    reference's (cfg-5's first step) -- no stale out-of-memory is reported by the launch checks that follow."""
    from pyfeaturetrack_amd.backend import Context, KltOutOfMemory
    frames, tc, n = baseline_case("cfg5")
    ctx = Context(0)
    pieces = []
    try:
        ctx.configure(tc)
        ctx.upload(0, frames[0])
        ctx.upload(1, frames[1])
        free0, total = ctx.device_memory()
        assert free0 > 8 * GB
        for size in (32 * GB, 4 * GB, 512 << 20, 64 << 20, 8 << 20):
            while len(pieces) < 4096:
                try:
                    pieces.append(ctx.device_alloc(size))
                except KltOutOfMemory as e:
                    assert "%d bytes asked for" % size in str(e), str(e)
                    break
        free1, _ = ctx.device_memory()
        assert free1 < 96 << 20, "%d MB still free after %d pieces" % (free1 >> 20, len(pieces))
        with pytest.raises(KltOutOfMemory) as ei:
            ctx.build_pyramids_batch([0, 1], sync=True)
        assert "bytes asked for" in str(ei.value)
        with pytest.raises(KltOutOfMemory):
            ctx.select(0, n, use_pyramid=False)                     # (the selection's own scratch does not fit either)
        assert not ctx.pyramids_valid(0)
        for p in pieces:
            ctx.device_free(p)
        pieces = []
        free2, _ = ctx.device_memory()
        assert free2 > free0 - GB
        ctx.build_pyramids_batch([0, 1], sync=True)                 # the very next calls: launch checks included (HIPCHK(hipGetLastError()))
        fl, placed = ctx.select(0, n, use_pyramid=True)
        assert placed == n
        _golden(fl, big, "cfg5", "sel")
        out, tracked = ctx.track(0, 1, fl)
        _golden(out, big, "cfg5", "trk")
        assert tracked == int((big["cfg5_trk_val"] >= 0).sum())
    finally:
        for p in pieces:
            ctx.device_free(p)
        ctx.close()
