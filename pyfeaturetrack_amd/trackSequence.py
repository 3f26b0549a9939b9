"""Device-resident sequence tracking: the call pattern of upstream KLT's example3 (select once; per frame
KLTTrackFeatures in sequentialMode, KLTReplaceLostFeatures, KLTStoreFeatureList) without a host round trip per frame.

The per-frame feature lists are the rows of ONE device-side [nFrames x nFeatures] record array (feature-buffer views,
include/klt_gpu.h klt_featbuf_view); row k-1 is the tracker's input and row k its output, replacement runs in place on
row k, and the whole table is downloaded once at the end into a KLT_FeatureTable.  Results
are bit-identical to the per-frame host API loop, which the GPU tests check.
"""
from __future__ import print_function

import numpy as np

from .backend import REPLACING_SOME, SELECTING_ALL, context_of
from .klt import KLT_FeatureTable
from .selectGoodFeatures import _fix_window, _slots_of

_FB_TABLE = 60000            # feature-buffer ids used by this module: the table, then one view per frame
_FB_ROW0 = 60001
_MAX_FRAMES = 65535 - _FB_ROW0
_OPT_SELECT_AFFINE_STATE = 4


_OPT_BUILD_STREAM = 15
_OPT_COPY_STREAMS = 20
SEQUENCE_COPY_STREAMS = 1        # copy streams a sequence call's uploads use (the default of a context, two, is for the two frames of a pair)


STAGER_WORKERS = 0                       # 0: by frame size (below); 1 / 2 / ...: that many helper threads
STAGER_THREADS_FROM_BYTES = 4 << 20      # frames of this size and above are staged by two helper threads (one core copies a 4K frame in
                                         # 0.26 ms -- longer than the GPU needs for it; tools/stage_copy_probe.py)


class _FrameStager:
    """Reads the frames and copies them into pinned staging buffers on helper threads (the copy and the frame source's decoding
    release the GIL), so that the host copy of frame k+2 -- 2 MB at 1080p, as long as the tracker of a frame runs -- is made while the
    calling thread waits for the GPU in frame k's replacement pass.  Every call into the library's device side stays on the calling
    thread.  With `workers` > 1 several frames are copied at the same time, each by one core: the iterator is advanced and the
    staging buffer taken under one lock, in frame order (so the buffers are handed out exactly as a single helper would), the copies
    run side by side, and `next()` delivers the frames in order."""

    def __init__(self, frames, buffers, shape, workers=1):
        import queue
        import threading
        self._frames, self._shape = frames, shape
        self._float = bool(buffers) and buffers[0].dtype == np.float32     # the staging buffers take float frames (colour / float images)
        self._free = queue.Queue()
        self._stop = threading.Event()
        self._pull = threading.Lock()               # advancing the iterator + taking a buffer: one frame at a time, in order
        self._cv = threading.Condition()            # results by frame index
        self._results, self._next_in, self._next_out, self._ended = {}, 0, 0, False
        for b in buffers:
            self._free.put(b)
        self._threads = [threading.Thread(target=self._run, name="klt-frame-stager-%d" % i, daemon=True) for i in range(max(1, workers))]
        for t in self._threads:
            t.start()

    def _publish(self, idx, kind, item):
        with self._cv:
            self._results[idx] = (kind, item)
            self._cv.notify_all()

    def _run(self):
        from ._abi import load_library
        from ._frames import FrameKey
        load_library().klt_host_thread_serial(1)            # (one core per frame: spread over the pool's lanes this background copy slowed the loop, profiles/README.md)
        while True:
            idx = None
            try:
                with self._pull:
                    if self._stop.is_set() or self._ended:  # closed (no further frame is pulled, no buffer written) / the source has ended
                        return
                    idx = self._next_in
                    try:
                        img = next(self._frames)
                    except StopIteration:
                        self._ended = True
                        self._publish(idx, None, None)
                        return
                    except BaseException as e:              # noqa: BLE001 -- the source failed: said at this frame's place, under the lock
                        self._ended = True                  # (another helper must not pull from the dead iterator and report the END here)
                        self._publish(idx, "error", e)
                        return
                    self._next_in = idx + 1
                    key = FrameKey(img)                     # (a Pillow image is read through its row table: no array is made of it)
                    kind = key.stage_kind()                 # ("u8" | "rgbx" | "f32", shape) of what can be written straight into a buffer
                    if kind is None and self._float and isinstance(img, np.ndarray) and img.dtype == np.float32 and img.ndim == 2:
                        kind = ("array", img.shape)
                    if kind is None or tuple(kind[1]) != tuple(self._shape) or (kind[0] == "u8") == self._float:
                        self._publish(idx, "raw", key.array())      # the calling thread deals with it (size error, or a synchronous upload)
                        continue
                    buf = self._free.get()
                    if buf is None or self._stop.is_set():
                        return
                if kind[0] == "rgbx":
                    key.float_into(buf)                     # Pillow's luma of a colour frame, straight into the pinned float buffer
                else:
                    key.copy_into(buf)
                self._publish(idx, "staged", buf)
            except BaseException as e:                      # noqa: BLE001 -- handed to the calling thread
                self._ended = True
                self._publish(self._next_in if idx is None else idx, "error", e)
                return

    def next(self):
        with self._cv:
            while self._next_out not in self._results:
                self._cv.wait()
            kind, item = self._results.pop(self._next_out)
            if kind is not None and kind != "error":
                self._next_out += 1                         # (the end and an error are final: asked again, they answer again)
            else:
                self._results[self._next_out] = (kind, item)
        if kind == "error":
            raise item
        return kind, item

    def release(self, buf):
        self._free.put(buf)

    def close(self):
        """Stops the helper threads: the stop flag is looked at before every frame and before every copy, the free queue is drained
        (so that no thread can pick up a buffer any more) and a sentinel per thread wakes one waiting for a buffer.  Returns True
        when every thread has ended -- only then may the staging buffers be handed to somebody else."""
        import queue
        self._stop.set()
        try:
            while True:
                self._free.get_nowait()
        except queue.Empty:
            pass
        for _ in self._threads:
            self._free.put(None)
        for t in self._threads:
            t.join(timeout=2.0)               # (a frame source that blocks keeps its daemon thread; it touches no buffer any more)
        return not any(t.is_alive() for t in self._threads)


def KLTTrackSequence(tc, frames, nFeatures, replace_lost=True, async_ingest=True, prefetch=True):
    """Track `nFeatures` features through `frames` (an iterable of equally sized 8-bit images) and return the
    KLT_FeatureTable: row 0 = the selected features, row k = the features after tracking frame k-1 -> k (and, with
    `replace_lost`, after replacing the lost ones on frame k: new features carry their eigenvalue in `val`, as after
    KLTReplaceLostFeatures).  tc.affineConsistencyCheck >= 0 runs the affine check on every step.
    `prefetch`: the pyramids of frame k+1 are built on a second HIP stream (KLT_OPT_BUILD_STREAM) while frame k is tracked and its
    lost features are replaced -- same results, the frames then live in a ring of three slots."""
    ctx = context_of(tc)
    with ctx.lock:                       # one KLT* call at a time per device context (backend.default_context)
        ctx.settle_deferred()
        return _track_sequence_locked(ctx, tc, frames, nFeatures, replace_lost, async_ingest, prefetch)


def _track_sequence_locked(ctx, tc, frames, nFeatures, replace_lost, async_ingest, prefetch):
    frames = iter(frames)
    _fix_window(tc)
    from ._frames import cache_of
    cache_of(tc).keep_all_handles()     # pyramid handles somebody kept (ComputeImagePyramids, tc.pyramid_last) fetch their planes first
    ctx.configure(tc)
    s = list(_slots_of(tc))              # ring of three frame slots (the third is otherwise the per-frame API's selection slot)
    cache_of(tc).forget()               # this call fills the slots itself: what the per-frame API remembers of them is void
    ring = 3 if prefetch else 2
    from ._frames import pixels_of
    first = pixels_of(next(frames))
    rows = [first]
    nrows, ncols = first.shape
    affine = tc.affineConsistencyCheck >= 0

    # frames are consumed lazily; the table grows in chunks so that a generator of unknown length works
    chunk = 64
    tables = []                         # [(fb_table, first_row, n_rows)]

    def row_fb(k):
        ci, off = divmod(k, chunk)
        if ci == len(tables):
            if ci * (chunk + 1) + chunk + 1 > _MAX_FRAMES:
                raise ValueError("sequence too long for one call")
            base = _FB_TABLE + ci * (chunk + 1)
            ctx.featbuf_alloc(base, chunk * nFeatures)
            for j in range(chunk):
                ctx.featbuf_view(base + 1 + j, base, j * nFeatures, nFeatures)
            tables.append(base)
        return tables[ci] + 1 + off

    workers = STAGER_WORKERS or (2 if first.nbytes >= STAGER_THREADS_FROM_BYTES else 1)
    # frames k+1 .. k+3 on their way, one being filled by each helper thread, one spare; uint8 buffers for 8-bit frames, float32 ones for
    # colour / float frames (the frame `img.convert("F")` would be, made by the helper threads: klt_upload_f32_async takes it from there)
    stage_key = ((nrows, ncols), ring + 2 + workers, first.dtype)
    stage = ctx.staging(*stage_key) if async_ingest and first.dtype in (np.uint8, np.float32) else None
    in_flight = []                          # staging buffer whose host-to-device copy may still be running

    def ingest(slot, img, k, staged=False, wait=True):
        if img.shape != (nrows, ncols):
            from .error import KLTError
            KLTError("(KLTTrackSequence) Size of incoming image ({0} by {1}) is different from size of previous image "
                     "({2} by {3})".format(img.shape[1], img.shape[0], ncols, nrows))
        if staged:
            if wait:
                ctx.upload_wait()           # the previous frame's copy has finished (kernels keep running): its buffer goes back
                while in_flight:
                    stager.release(in_flight.pop()[1])
            ctx.upload_async(slot, img)
            in_flight.append((k, img))
        else:
            ctx.upload(slot, img)

    def landed(k):
        """frame k's selection has completed, so its pyramid was built, so its copy has left the staging buffer: the buffer goes back
        (no host wait for the copy streams -- klt_upload_wait costs the host the rest of the copy)"""
        while in_flight and in_flight[0][0] <= k:
            stager.release(in_flight.pop(0)[1])

    ingest(s[0], first, 0)
    ctx.build_pyramids(s[0], sync=False)
    # the initial selection follows KLTSelectGoodFeatures: the smoothed level-0 image when tc.smoothBeforeSelecting, the raw
    # frame otherwise (selectGoodFeatures.py:183-197) -- level 0 of the pyramid is always smoothed
    ctx.select_async(s[0], SELECTING_ALL, bool(tc.smoothBeforeSelecting), row_fb(0), nFeatures)
    state = None
    if affine:
        state = ctx.take_affine_state()
        ctx.affine_alloc(state, nFeatures)
    k = 0
    stager = _FrameStager(frames, stage, (nrows, ncols), workers=workers) if stage is not None else None
    try:
        if affine:
            ctx.set_option(_OPT_SELECT_AFFINE_STATE, state)
        if prefetch:
            ctx.set_option(_OPT_BUILD_STREAM, 1)
        # one new frame per step: all uploads on ONE copy stream (the default, two, is for the two frames of a pair; with two, the look at
        # every other frame's selection waits twice as long -- 0.280 against 0.307 ms per 4K frame, profiles/README.md round 6)
        ctx.set_option(_OPT_COPY_STREAMS, SEQUENCE_COPY_STREAMS)

        def stage_frame(k, item):                    # upload + pyramid build of frame k (enqueued only)
            ingest(s[k % ring], item[1], k, staged=item[0] == "staged")
            ctx.build_pyramids(s[k % ring], sync=False)
            if replace_lost and prefetch:            # ... and the list-independent half of its replacement pass, on the same stream
                ctx.select_prepare(s[k % ring])

        def next_frame():                            # ("staged", pinned buffer) | ("raw", array) | None at the end
            if stager is not None:
                item = stager.next()
                return item if item[0] is not None else None
            img = next(frames, None)
            return None if img is None else ("raw", pixels_of(img))

        def track(j):                                # frame j - 1 -> j: row j - 1 of the table in, row j out
                cur, prev = s[j % ring], s[(j - 1) % ring]
                if affine:
                    ctx.track_affine_async(prev, cur, row_fb(j - 1), row_fb(j), nFeatures, state)
                else:
                    ctx.track_async(prev, cur, row_fb(j - 1), row_fb(j), nFeatures)

        # The tracker of frame k + 1 only READS the list that frame k's replacement completes, so it is enqueued before the host looks at
        # that replacement's outcome (klt_select_finish): the GPU has it queued while the host turns around.  In the rare case that the
        # look makes the selection rewrite the list, the tracker is enqueued once more.  Not with the affine check: it updates the
        # per-feature state in place, so its launch cannot simply be repeated.
        ahead = prefetch and replace_lost and not affine
        if ahead:
            # Frames are SENT two steps before their pyramid is built: frame k + 3 leaves for the slot of frame k (its other raw buffer) at the
            # end of step k, while the tracker k -> k + 1 still reads that slot's pyramids.  Sent one step ahead, a 4K frame's 155 us copy and
            # its build sit behind each other in front of the tracker that needs them (0.329 instead of 0.277 ms per frame, profiles/README.md).
            def send(j):
                item = next_frame()
                if item is not None:
                    ingest(s[j % ring], item[1], j, staged=item[0] == "staged", wait=False)
                return item is not None

            def build(j):
                ctx.build_pyramids(s[j % ring], sync=False)
                ctx.select_prepare(s[j % ring])

            have = 0                                 # frames sent so far beyond frame 0: 1 .. have
            for j in (1, 2):
                if have == j - 1 and send(j):        # (a source that has ended is not asked again: the helper thread says so only once)
                    have = j
            if have >= 1:
                build(1)
                track(1)
            if have == 2 and send(3):                # into frame 0's slot, which only the tracker just enqueued still reads
                have = 3
            while k + 1 <= have:
                k += 1
                ctx.select_begin(s[k % ring], REPLACING_SOME, True, row_fb(k), nFeatures)
                if k + 1 <= have:
                    build(k + 1)
                    track(k + 1)
                if ctx.select_finish() and k + 1 <= have:
                    track(k + 1)
                if stager is not None:
                    landed(k)
                # (after the look: a repeated tracker needs slot k's pyramids valid.  Putting the send off to behind the NEXT selection's
                # launches -- so that the host is back at them sooner -- measured slower, 0.360 against 0.335 ms per 4K frame, 0.168 against
                # 0.153 at 1080p: the copy's head start is worth more than the host's 25 us)
                if have == k + 2 and send(k + 3):
                    have = k + 3
            nxt = None
        else:
            nxt = next_frame()
            if nxt is not None:
                stage_frame(1, nxt)
        while nxt is not None:
            k += 1
            nxt = next_frame()
            if not ahead:
                track(k)
            if replace_lost:                         # the chain first: tracker, then the replacement pass up to the host's look at it
                ctx.select_begin(s[k % ring], REPLACING_SOME, True, row_fb(k), nFeatures)
            if nxt is not None and prefetch:
                stage_frame(k + 1, nxt)              # the next frame's upload, build and scores: enqueued while the GPU works on the above
                if ahead:
                    track(k + 1)
            if replace_lost:
                redo = ctx.select_finish()
                if redo and ahead and nxt is not None:
                    track(k + 1)
            if nxt is not None and not prefetch:
                stage_frame(k + 1, nxt)
    finally:
        # every step on its own: a failing one must not skip the rest (the options live on the shared default context)
        try:
            ctx.select_finish()                      # (only pending after an exception)
        finally:
            try:
                if stager is not None and not stager.close():
                    # the helper thread is still inside the caller's frame source: it may yet write one buffer of this set, so the
                    # next call must not get the same pinned memory (a fresh set is allocated instead)
                    ctx.staging_forget(*stage_key)
            finally:
                try:
                    ctx.set_option(_OPT_COPY_STREAMS, 2)
                    if prefetch:
                        ctx.set_option(_OPT_BUILD_STREAM, 0)
                finally:
                    if affine:
                        try:
                            ctx.set_option(_OPT_SELECT_AFFINE_STATE, -1)
                        finally:
                            ctx.release_affine_state(state)
    nframes = k + 1
    ft = KLT_FeatureTable(nframes, nFeatures, _fill=False)
    for ci, base in enumerate(tables):
        lo = ci * chunk
        hi = min(nframes, lo + chunk)
        ctx.featbuf_download_into(base, ft.rec[lo:hi])          # rows lo .. hi-1 are contiguous in the table: no staging copy
    ft.rec["aux"] = 0
    if tc.sequentialMode:
        # leave the context as the per-frame API would: the last frame's pyramids are "frame 1" of the next call
        from .trackFeatures import _pyramid_handles
        last = s[(nframes - 1) % ring]
        if last != s[0]:
            ctx.swap_slots(s[0], last)
        tc.pyramid_last, tc.pyramid_last_gradx, tc.pyramid_last_grady = _pyramid_handles(tc, ctx, s[0], ncols, nrows)
    return ft
