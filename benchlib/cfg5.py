"""bench.py --config cfg5: one 4K sequence with replacement after every frame (one GPU: the whole clip; N GPUs: blocks of frames, the list as a baton)."""
from .common import *            # noqa: F401,F403 -- the shared helpers, constants and the modules they import (np, os, time, ...)


def run_cfg5(args, json_fd):
    """BASELINE cfg-5 (single GPU): 3840x2160 sequence, 20000 features, sequential mode, lost features replaced after every
    frame.  Per step (= frame): upload is excluded (frames resident), pyramid of the new frame, track, REPLACING_SOME selection."""
    ranks = Ranks(args)
    if ranks.distributed:
        return run_cfg5_blocks(args, json_fd, ranks)
    w, h, n = 3840, 2160, 20000
    nframes = max(2, args.frames)                # BASELINE cfg-5: a 512-frame sequence; every frame resident in its own slot (115 MB of
    tc = cfg2_context()                          # raw frame + pyramid planes each: 59 GB of the 288 GB for 512 frames)
    tc.max_residue = 10.0
    p = params_from_tc(tc)
    ctx = Context(0)
    ctx.configure(tc)
    phases = synth.sequence_phases(w, h, 4, workers=usable_cores(10))
    # the clip is resident in HBM as u8 (4.2 GB for 512 frames); a frame's slot ADOPTS its buffer (klt_slot_adopt_u8: read in place, no
    # copy), the slots are a ring of three as in any sequence -- one slot per frame (59 GB of planes, each written once per pass) reads
    # 0.336 ms per frame instead of 0.26: fresh pages for 115 MB of planes every frame
    NPX = w * h
    store = ctx.device_alloc(nframes * NPX)
    frames = []                                  # only the first frames stay on the host (parity check, CPU baseline)
    for k, f in enumerate(synth.periodic_sequence(w, h, 4, nframes, phases=phases)):
        ctx.device_write(store + k * NPX, f)
        if k < 8:
            frames.append(f)
    RING = (10, 11, 12)

    def slot(k):
        return RING[k % 3]

    def build(k, prepare_scores=False):
        ctx.adopt_u8(slot(k), store + k * NPX, w, h)
        ctx.build_pyramids(slot(k), sync=False)
        if prepare_scores:
            ctx.select_prepare(slot(k))

    build(0)
    fl, placed = ctx.select(slot(0), n, use_pyramid=True)
    ctx.featbuf_upload(0, fl)
    ctx.sync()

    # the pyramids of frame k+1 are built on the context's build stream while frame k is tracked and its lost features are replaced
    # (KLT_OPT_BUILD_STREAM; same results -- every frame has its own slot here)
    prefetch = os.environ.get("KLT_BENCH_NO_PREFETCH") != "1"
    prepare = prefetch and os.environ.get("KLT_BENCH_NO_PREPARE") != "1"
    if prefetch:
        ctx.set_option(15, 1)

    redone = [0]

    def run_sequence(look=None):
        """one pass over the sequence; `look(k)` (instrumented passes) is called after frame k's replacement, synchronised"""
        def track(k):                                     # frame k - 1 -> k; the lists alternate between two buffers
            ctx.track_async(slot(k - 1), slot(k), (k - 1) % 2, k % 2, n)

        build(0)                                          # (a pass starts from frame 0 again: its slot holds a later frame by now)
        if prefetch:
            build(1, prepare)
            track(1)
        for k in range(1, nframes):
            if not prefetch:
                build(k)
                track(k)
            ctx.select_begin(slot(k), 2, True, k % 2, n)      # KLTReplaceLostFeatures on the resident level-0 images, up to the host's look
            if prefetch and k + 1 < nframes:
                build(k + 1, prepare)                     # pyramids + SAT + eigenvalues of the next frame, on the build stream
                # the NEXT frame's tracker goes out before the host looks at this frame's selection: it only reads the list, and the GPU
                # has it queued while the host turns around (44 us of an idle main stream per frame in the round-3 kernel trace)
                track(k + 1)
            if ctx.select_finish() and prefetch and k + 1 < nframes:
                redone[0] += 1
                track(k + 1)                              # (rare) the selection rewrote the list after the tracker had read it
            if look:
                ctx.sync()
                look(k)
        ctx.sync()

    def region():
        for _ in range(max(1, args.steps // (nframes - 1))):
            ctx.featbuf_upload(0, fl)
            run_sequence()

    frames_per_region = max(1, args.steps // (nframes - 1)) * (nframes - 1)
    ctx.featbuf_upload(0, fl)
    run_sequence()
    el, regions, enq = timed_regions(OneGpu([ctx]), region, max(5, min(args.repeats, 10)))
    out = ctx.featbuf_download((nframes - 1) % 2, n)

    # instrumented pass 1: the list after every frame (parity) and how long the replacement alone takes
    lists = {}
    ctx.featbuf_upload(0, fl)
    run_sequence(look=lambda k: lists.__setitem__(k, ctx.featbuf_download(k % 2, n)))
    same_end = bool(np.array_equal(lists[nframes - 1], out))
    t_sel, lost = 0.0, []
    ctx.featbuf_upload(0, fl)
    build(0)
    for k in range(1, nframes):                              # (plain loop: tracker, look at the losses, replacement timed on its own)
        build(k)
        ctx.track_async(slot(k - 1), slot(k), (k - 1) % 2, k % 2, n)
        lost.append(int((ctx.featbuf_download(k % 2, n)["val"] < 0).sum()))
        t = time.perf_counter()
        ctx.select_async(slot(k), 2, True, k % 2, n)
        ctx.sync()
        t_sel += time.perf_counter() - t
    ko = load_oracle()
    checks = [("the timed passes end with the list of the instrumented pass", same_end, 0.0)]
    PAR_FRAMES = 3
    if ko:
        ko.set_threads(usable_cores())
        ofl = ko.select_good_features(p, frames[0].astype(np.float32), n)
        checks.append(("selection of 20000 on frame 0", bool(np.array_equal(ofl["x"], fl["x"]) and np.array_equal(ofl["y"], fl["y"]) and np.array_equal(ofl["val"], fl["val"])), 0.0))
        P_prev = ko.Pyramids(p, frames[0].astype(np.float32))
        for k in range(1, PAR_FRAMES + 1):
            P_cur = ko.Pyramids(p, frames[k].astype(np.float32))
            ko.track_features(p, P_prev, P_cur, ofl)
            ofl = ko.select_good_features(p, frames[k].astype(np.float32), n, mode=2, fl=ofl)
            same, dx = records_equal(lists[k], ofl)
            checks.append(("list after tracking into frame %d and replacing the lost features" % k, same, dx))
            P_prev = P_cur
        ko.set_threads(1)
    par = parity_summary(checks, "initial selection and the whole feature list (tracked and replaced records) after each of the first %d frames; "
                         "the wrapper KLTReplaceLostFeatures is absent from the reference (pinned at the level of _enforceMinimumDistance)" % PAR_FRAMES)

    # instrumented pass 2: every launch timed (one stream order per stream; the build stream's launches carry their own timestamps)
    def seq():
        ctx.featbuf_upload(0, fl)
        run_sequence()

    seq()
    ctx.track_stats_reset()
    paired = timed_pass(ctx, seq, 1)
    st = ctx.track_stats()
    sane_iterations(st, st["features"], p.nPyramidLevels, "cfg-5")
    stamped = timed_pass(ctx, seq, 2)
    ms_step = el / frames_per_region * 1e3
    kt = kernel_table(stamped, paired, nframes - 1, {"track": track_bytes(p, st, (nframes - 1) * n)})
    roof = roofline_of(kt, nframes - 1, ms_step, extra={"newton_iterations_per_level": [v / (nframes - 1.0) for v in st["iterations"][:p.nPyramidLevels]]})
    cpu = None
    if ko and not args.no_cpu_baseline:
        a0, a1 = frames[0].astype(np.float32), frames[1].astype(np.float32)
        P0 = ko.Pyramids(p, a0)

        def one_frame():
            o = fl.copy()
            ko.track_features(p, P0, ko.Pyramids(p, a1), o)
            ko.select_good_features(p, a1, n, mode=2, fl=o)

        cpu = cpu_baseline_of(ko, one_frame, n, "pyramid of the new 3840x2160 frame + track 20000 features + replacement selection (one frame of cfg-5)", budget_s=8.0)
    ctx.close()
    line = base_line(n * frames_per_region / el, 1, frames_per_region, 0, ms_step, ms_step,
                     "cfg-5 (one GPU): ONE 3840x2160 sequence of %d frames (the clip resident in HBM as u8; a ring of three frame slots adopts the frames in place), 20000 features, 7x7, 3 levels (ss 4), sequential mode, lost " % nframes +
                     "features replaced after every frame; per step (frame): pyramid of the new frame + track + replacement"
                     + ("; the next frame's pyramids are built on a second stream meanwhile" if prefetch else "")
                     + (", and so are its summed-area tables and eigenvalues (klt_select_prepare_async)" if prepare else ""),
                     extra_cfg={"frames": nframes, "live_at_end": int((out["val"] >= 0).sum()), "ms_replace_per_frame": t_sel / (nframes - 1) * 1e3,
                                "lost_per_frame": {"first": lost[:8], "min": min(lost), "median": float(statistics.median(lost)), "max": max(lost)},
                                "build_stream": bool(prefetch), "scores_prepared": bool(prepare),
                                "tracker_enqueued_ahead": bool(prefetch), "trackers_repeated": redone[0]})
    line.update(par)
    line["roofline"], line["cpu_baseline"] = roof, cpu
    line["extra"] = {"region_ms_per_step": region_stats(regions, frames_per_region, el), "host_enqueue_ms_per_step": enq / frames_per_region * 1e3}
    emit(json_fd, line)
    fail_on_parity(par)


def run_cfg5_blocks(args, json_fd, ranks):
    """cfg-5 on N GPUs: ONE 3840x2160 sequence cut into blocks of 7 tracking steps, block r on rank r.  The tracker and
    the replacement pass are a serial chain, so the feature list travels from rank to rank as a baton (klt_sendrecv_featbuf_async,
    320 KB); what depends on the pixels only -- the pyramids and the selection scores of the block's frames -- is enqueued on the
    owner's build stream at once and is ready (ranks > 0) long before the baton arrives.  Per-GPU work is fixed: weak scaling; the
    serial chain bounds it.  With one rank (KLT_FORCE_DIST=1) this is the single-GPU sequence with all pixel
    work enqueued ahead, and the baton a device copy."""
    w, h, n, B = 3840, 2160, 20000, 7
    rank, world = ranks.rank, ranks.world
    tc = cfg2_context()
    tc.max_residue = 10.0
    ctx = Context(ranks.local_rank)
    ctx.configure(tc)
    base = synth.synth_base(w, h, 4)
    first = rank * B                                       # global index of the block's frame 0 (= the previous block's last frame)
    for j in range(B + 1):
        ctx.upload(10 + j, synth.synth_frame(w, h, 4, first + j, base=base))
    fl = None
    if rank == 0:
        ctx.build_pyramids(10)
        fl, placed = ctx.select(10, n, use_pyramid=True)
    ctx.set_option(15, 1)                                  # KLT_OPT_BUILD_STREAM
    ctx.set_option(16, B + 1)                              # KLT_OPT_SCORE_SETS: one per frame of the block
    ranks.attach([ctx])
    FB_A, FB_B, FB_BATON, FB_ALL = 0, 1, 2, 3

    def block():
        ctx.comm_fence_featbuf(FB_B if B % 2 else FB_A)    # the baton sent at the end of the previous block has left its buffer
        for j in range(B + 1):                             # the block's pixel work: build stream, nothing to wait for
            ctx.build_pyramids(10 + j, sync=False)
            if j:
                ctx.select_prepare(10 + j)
        if rank == 0:
            ctx.featbuf_upload(FB_A, fl)
        else:
            ctx.sendrecv_featbuf(-1, -1, FB_A, rank - 1, n)             # the baton: the list after the previous block's last frame
        for j in range(1, B + 1):
            ctx.track_async(10 + j - 1, 10 + j, (FB_A, FB_B)[(j - 1) % 2], (FB_A, FB_B)[j % 2], n)
            ctx.select_async(10 + j, 2, True, (FB_A, FB_B)[j % 2], n)        # (nothing to enqueue in between: the pixel work is ahead)
        last = (FB_A, FB_B)[B % 2]
        if world > 1 and rank + 1 < world:
            ctx.sendrecv_featbuf(last, rank + 1, -1, -1, n)
        elif world == 1:
            ctx.sendrecv_featbuf(last, 0, FB_BATON, 0, n)                # one rank: the baton path as a device copy
        return last

    last = block()                                         # warm-up (allocations, RCCL's lazy connections)
    ranks.fence()
    # a timed region is ONE pass of the sequence over the ranks (a second pass inside the region would let rank 0 start it while the
    # others still work on the first: N pipelined replicas, not one sequence)
    reps = 1
    el, regions, enq = timed_regions(ranks, block, max(5, min(args.repeats, 15)))
    rccl = ranks.validation(1)
    # the list after the last frame of every block, gathered on rank 0 (rank order = frame order)
    ctx.gather_featbuf_async(last, FB_ALL, n, 0)
    ctx.comm_wait()
    ctx.sync()
    if rank == 0:
        table = ctx.featbuf_download(FB_ALL, n * world).reshape(world, n)
        baton_ok = None
        if world == 1:
            baton_ok = bool(np.array_equal(ctx.featbuf_download(FB_BATON, n), table[0]))
        frames_done = reps * B * world
        emit(json_fd, base_line(n * frames_done / el, world, frames_done, 0, el / frames_done * 1e3, el / (reps * B) * 1e3,
                                "cfg-5 on %d GPU(s): ONE 3840x2160 sequence in blocks of %d frames per GPU, 20000 features, sequential "
                                "mode, lost features replaced after every frame; the feature list is the baton between the blocks (RCCL "
                                "send / receive), the blocks' pyramids and selection scores are prepared on the owners' build streams" % (world, B),
                                extra_cfg={"rccl_ranks": rccl.get("rccl_ranks", 0), "rccl_validation": rccl, "live_after_each_block": [int((t["val"] >= 0).sum()) for t in table],
                                           "list_sha16_after_each_block": [list_digest(t) for t in table],
                                           "ms_per_frame_of_the_chain": el / (reps * B * world) * 1e3, "baton_copy_ok": baton_ok,
                                           "region_ms": {"median": el * 1e3, "min": min(regions) * 1e3, "max": max(regions) * 1e3}}))
    ctx.close()
    Ranks.fail_on_validation(rccl)

