"""bench.py (default) cfg-2, the headline: distinct resident 1080p pairs, batched launches on several contexts; secondary figures (ingest, sequences from host memory, the reference-shaped Python API)."""
from .common import *            # noqa: F401,F403 -- the shared helpers, constants and the modules they import (np, os, time, ...)


T_OUT0, T_OUT1, T_GATH0, T_GATH1, FB_IN0, V_OUT0, FB_MISC = 10, 11, 20, 21, 1000, 3000, 90


def run_cfg2(args, json_fd):
    ranks = Ranks(args)
    rank, world, distributed = ranks.rank, ranks.world, ranks.distributed
    tc = cfg2_context()
    p = params_from_tc(tc)
    nctx, B, NP = max(1, args.inflight), max(1, args.batch), args.resident_pairs
    if NP < nctx * B or NP % (nctx * B):
        raise SystemExit("--resident-pairs must be a positive multiple of --inflight x --batch")
    PL = NP // nctx                              # pairs per context
    NG = PL // B                                 # groups (launch sets) per context and step
    seeds = [rank * NP + i + 1 for i in range(NP)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, usable_cores(16) // max(1, min(world, 8)))) as ex:
        frames = list(ex.map(lambda s: synth.synth_pair(WIDTH, HEIGHT, seed=s), seeds))

    # pair i: group i // B of the step; groups go round-robin to the contexts.  Context c, its j-th group, pair b of the group:
    # local pair index lp = j B + b, frame slots 2 lp and 2 lp + 1, input list FB_IN0 + lp, output row lp of the step's table
    def pair_index(c, lp):
        j, b = divmod(lp, B)
        return (j * nctx + c) * B + b

    ctxs, lists = [], {}
    for c in range(nctx):
        cx = Context(ranks.local_rank)
        cx.set_params(p)
        for lp in range(PL):
            f0, f1 = frames[pair_index(c, lp)]
            cx.upload(2 * lp, f0)
            cx.upload(2 * lp + 1, f1)
        for t in (T_OUT0, T_OUT1):
            cx.featbuf_alloc(t, PL * NFEAT)
        for t in (0, 1):
            for lp in range(PL):
                cx.featbuf_view(V_OUT0 + t * PL + lp, (T_OUT0, T_OUT1)[t], lp * NFEAT, NFEAT)
        for j in range(PL // B):                 # (launches of the same shape as the timed ones: a profiler's per-kernel averages stay clean)
            cx.build_pyramids_batch([2 * (j * B + b) + f for b in range(B) for f in (0, 1)])
        for lp in range(PL):
            fl_c, placed = cx.select(2 * lp, NFEAT, use_pyramid=True)
            assert placed == NFEAT, "only %d of %d features could be placed" % (placed, NFEAT)
            lists[pair_index(c, lp)] = fl_c
            cx.featbuf_upload(FB_IN0 + lp, fl_c)
        ctxs.append(cx)
    ctx = ctxs[0]
    ranks.attach(ctxs)

    def group_slots(j, nb=B):
        return [2 * (j * B + b) + f for b in range(nb) for f in (0, 1)]

    def group_build(cx, j):
        cx.build_pyramids_batch(group_slots(j))                  # all frames of the group share every launch

    def group_track(cx, j, t):
        if B == 1:
            cx.track_async(2 * j, 2 * j + 1, FB_IN0 + j, V_OUT0 + t * PL + j, NFEAT)
        else:
            cx.track_batch_async([(2 * lp, 2 * lp + 1, FB_IN0 + lp, V_OUT0 + t * PL + lp) for lp in range(j * B, j * B + B)], NFEAT)

    step_no = [0]

    def one_step(collect=True):
        """one pass over the resident pairs (collect = False: without the collectives -- the untimed prewarm runs for a TIME, i.e. a
        different number of steps on every rank, and a collective must be issued by all ranks or by none).  The groups go out in rounds of one group per context, the builds of a round before its
        tracker launches: every stream has work a few microseconds after the step starts (enqueueing a group takes the host ~25 us);
        the order inside each stream, and the work, are the same either way.  N > 1: the step's record table of every context is
        all-gathered with ONE collective behind its last tracker launch; two tables alternate, a table is reused once its collective
        of two steps ago has read it."""
        t = step_no[0] % 2
        step_no[0] += 1
        gather = distributed and collect
        if distributed:
            for cx in ctxs:
                cx.comm_fence_featbuf((T_OUT0, T_OUT1)[t])
        for j in range(NG):
            for cx in ctxs:
                group_build(cx, j)
            for cx in ctxs:
                group_track(cx, j, t)
        if gather:
            for cx in ctxs:
                cx.allgather_featbuf_async((T_OUT0, T_OUT1)[t], (T_GATH0, T_GATH1)[t], PL * NFEAT)
        return t

    # bring the GPU to its steady state first (the same work as the steps)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        one_step(collect=False)
        for cx in ctxs:
            cx.sync()
    for _ in range(args.warmup):
        one_step()
    if os.environ.get("KLT_BENCH_DIE_RANK") == str(rank):      # test hook: a rank that vanishes with collectives in flight
        os._exit(7)

    def region():
        for _ in range(args.steps):
            one_step()

    elapsed, regions, enqueue_s = timed_regions(ranks, region, args.repeats)
    t_last = (step_no[0] - 1) % 2
    rccl = ranks.validation(args.steps)                 # (collective: every rank, right behind the timed regions)

    # correctness of what was timed: the last step's records of EVERY resident pair (and, N > 1, what the gather delivered of them)
    outs = {}
    for c, cx in enumerate(ctxs):
        tab = cx.featbuf_download((T_OUT0, T_OUT1)[t_last], PL * NFEAT).reshape(PL, NFEAT)
        for lp in range(PL):
            outs[pair_index(c, lp)] = tab[lp]
        if distributed:                 # what this rank received from itself equals what it produced
            got = cx.featbuf_download((T_GATH0, T_GATH1)[t_last], world * PL * NFEAT).reshape(world, PL, NFEAT)
            assert np.array_equal(got[rank], tab), "gathered records differ"
    out, fl = outs[0], lists[0]
    tracked = int(np.count_nonzero(out["val"] >= 0))
    live = out["val"] == 0
    shift = (float(np.median(out["x"][live] - fl["x"][live])), float(np.median(out["y"][live] - fl["y"][live])))
    ko = load_oracle() if rank == 0 else None
    parity = {}
    if rank == 0:
        checks = []
        if ko:
            nthreads = usable_cores()
            for i in range(NP):
                same, dx = records_equal(outs[i], oracle_track(ko, p, frames[i][0], frames[i][1], lists[i], threads=nthreads))
                checks.append(("pair %d (seed %d)" % (i, seeds[i]), same, dx))
        what = "tracked records of all %d resident pairs, last timed step" % NP
        if ko and world > 1:
            # ... and what another rank contributed: pair 0 of the last rank (context 0, row 0 of its table) as this rank received it,
            # against the oracle's selection + tracking of that pair from its seed
            g0, g1 = synth.synth_pair(WIDTH, HEIGHT, seed=(world - 1) * NP + 1)
            ko.set_threads(usable_cores())
            osel = ko.select_good_features(p, g0.astype(np.float32), NFEAT)
            ko.set_threads(1)
            got0 = ctxs[0].featbuf_download((T_GATH0, T_GATH1)[t_last], world * PL * NFEAT).reshape(world, PL, NFEAT)
            same, dx = records_equal(got0[world - 1][0], oracle_track(ko, p, g0, g1, osel, threads=usable_cores()))
            checks.append(("pair 0 of rank %d as gathered" % (world - 1), same, dx))
            what += " + pair 0 of the last rank as received through the all-gather"
        parity = parity_summary(checks, what)

    # second pass: per-kernel timing + iteration counters for the roofline, on context 0 over its own pairs
    roofline = None
    ms_per_pair = elapsed / (args.steps * NP) * 1e3
    if rank == 0:
        passes = max(1, min(args.steps, 4))

        def ctx0_passes(n=passes):
            for _ in range(n):
                for j in range(NG):
                    group_build(ctx, j)
                    group_track(ctx, j, 0)

        def warm():
            """at the clocks the timed regions ran at: the parity check and the downloads above left the GPU idle"""
            t_warm = time.perf_counter()
            while (time.perf_counter() - t_warm) * 1e3 < min(args.prewarm_ms, 30.0):
                ctx0_passes(1)
                ctx.sync()

        warm()
        ctx.track_stats_reset()                        # AFTER the warm-up: the counters cover exactly the launches they are divided by
        paired = timed_pass(ctx, ctx0_passes, 1)
        st = ctx.track_stats()
        npairs_roof = passes * PL
        sane_iterations(st, npairs_roof * NFEAT, p.nPyramidLevels, "cfg-2")
        warm()
        stamped = timed_pass(ctx, ctx0_passes, 2)
        kt = kernel_table(stamped, paired, npairs_roof, {"track": track_bytes(p, st, npairs_roof * NFEAT)})
        st_pair = {k: ([x / npairs_roof for x in v] if isinstance(v, list) else v / npairs_roof) for k, v in st.items()}
        pyr_b, trk_b = algorithmic_bytes(p, WIDTH, HEIGHT, st_pair, NFEAT)
        dom = "smooth_grad_l0"
        # PMC-derived figures are NOT measured by this run: committed results of the builder's rocprofv3 --pmc passes, with their
        # provenance, dropped when the kernel source changed since (committed_counters)
        traffic, traffic_source = committed_counters("traffic.json", dom, B)
        # the same kernel against the roof that actually bounds it: VALU issue.  Wavefront-instructions per launch come from a
        # rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/sq_counters.json, tools/pmc_sq.py); 4.5 clocks per instruction and SIMD
        # is what the FP64-rate instruction mix of the convolutions sustains on gfx950 (tools/mb/valu_rate.hip, fp64_mix.hip).
        issue = None
        sq, sq_source = committed_counters("sq_counters.json", dom, B)
        if sq and sq.get("SQ_INSTS_VALU"):
            simds, cpi, mhz = 256 * 4, 4.5, 2400.0
            ideal_us = sq["SQ_INSTS_VALU"] / simds * cpi / mhz
            issue = {"valu_wavefront_instructions_per_launch": sq["SQ_INSTS_VALU"], "simds": simds, "clocks_per_instruction": cpi,
                     "clock_mhz": mhz, "ideal_us": ideal_us, "frac": ideal_us / kt[dom]["us_per_launch"], "source": sq_source}
            if issue["frac"] > 1.0:
                # a MODEL (calibrated clocks per instruction x a committed instruction count), not a measurement of this run: when the
                # kernel beats it, the model is what is wrong -- say so instead of printing a fraction above 1
                issue["model_exceeded"] = issue.pop("frac")
        elif sq_source:
            issue = {"source": sq_source}
        npx = WIDTH * HEIGHT * 2 * B
        moved = npx * (1 + 4 + 12) + npx // p.subsampling * 4      # what crosses L2: u8 in, image + two gradients + the H1 plane out
        roofline = roofline_of(kt, npairs_roof, ms_per_pair, dominant=dom, extra={
            "traffic": traffic, "traffic_source": traffic_source, "issue_bound": issue, "pairs_per_launch": B,
            "frac_note": "frac books SURVEY 8(d)'s 21 B per pixel (17 for smoothing + gradients, 4 for the first reduction's input, which this "
                         "kernel consumes from LDS); frac_moved books the 18 B per pixel that actually cross the L2 (4 of the 21 never leave LDS, "
                         "the H1 plane adds 1)",
            "moved_bytes_per_launch": moved, "achieved_moved": moved / (kt[dom]["us_per_launch"] * 1e-6) / 1e9,
            "frac_moved": moved / (kt[dom]["us_per_launch"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "step_unit": "one frame pair", "step_algorithmic_bytes_formula": 2 * pyr_b + trk_b,
            "newton_iterations_per_level": st_pair["iterations"][:p.nPyramidLevels]})

    tree = None
    if rank == 0 and roofline and not args.no_extras:
        tree = tree_sums_probe(ctx, lambda: [group_track(ctx, j, 0) for j in range(NG)],
                               lambda: ctx.featbuf_download(T_OUT0, PL * NFEAT),
                               roofline["kernels"]["track"]["algorithmic_bytes_per_launch"], p.window_width)

    # secondary figures (never `value`): selection time, the one-stream figure, the cache-resident figure and the PCIe-inclusive pair time
    extra = None
    ms_single = None
    reg = region_stats(regions, args.steps, elapsed)
    if rank == 0 and args.no_extras:
        extra = {"region_ms_per_step": reg, "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3, "note": "--no-extras: secondary figures skipped"}
    elif rank == 0:
        reps = 10
        ctx.sync()
        t = time.perf_counter()
        for k in range(reps):
            ctx.select_async(2 * (k % PL), 1, True, FB_MISC, NFEAT)      # SELECTING_ALL on a resident level-0 pyramid
        ctx.sync()
        ms_select = (time.perf_counter() - t) / reps * 1e3
        def one_pair_at_a_time():
            runs = []
            for _ in range(5):
                t = time.perf_counter()
                for i in range(4 * PL):                          # one pair per build / tracker call, rotating through the context's pairs
                    lp = i % PL
                    ctx.build_pyramids_batch([2 * lp, 2 * lp + 1])
                    ctx.track_async(2 * lp, 2 * lp + 1, FB_IN0 + lp, V_OUT0 + lp, NFEAT)
                ctx.sync()
                runs.append((time.perf_counter() - t) / (4 * PL) * 1e3)
            return runs

        singles = one_pair_at_a_time()
        ms_single = statistics.median(singles)
        # round 2's headline arrangement: every context rebuilds the SAME two pairs (four slots) over and over, so the pyramid planes
        # the tracker reads are still in the Infinity Cache
        nrep, HB = 64, min(2, PL)                  # (two pairs per launch: 4 x 27 MB of planes per context stay below the cache's 256 MB)
        hot_pairs = [(2 * lp, 2 * lp + 1, FB_IN0 + lp, V_OUT0 + lp) for lp in range(HB)]
        hots = []
        for _ in range(5):
            for cx in ctxs:
                cx.sync()
            t = time.perf_counter()
            for _ in range(nrep):
                for cx in ctxs:
                    cx.build_pyramids_batch(list(range(2 * HB)))
                for cx in ctxs:
                    cx.track_batch_async(hot_pairs, NFEAT) if HB > 1 else cx.track_async(0, 1, FB_IN0, V_OUT0, NFEAT)
            for cx in ctxs:
                cx.sync()
            hots.append((time.perf_counter() - t) / (nrep * nctx * HB) * 1e3)
        ms_hot = statistics.median(hots)
        t = time.perf_counter()
        for k in range(reps):                                  # un-pipelined latency of one pair
            lp = k % PL
            ctx.build_pyramids_batch([2 * lp, 2 * lp + 1])
            ctx.track_async(2 * lp, 2 * lp + 1, FB_IN0 + lp, V_OUT0 + lp, NFEAT)
            ctx.sync()
        ms_latency = (time.perf_counter() - t) / reps * 1e3
        t = time.perf_counter()
        for k in range(reps):
            lp = k % PL
            f0, f1 = frames[pair_index(0, lp)]
            ctx.upload(2 * lp, f0)
            ctx.upload(2 * lp + 1, f1)
            ctx.build_pyramids_batch([2 * lp, 2 * lp + 1])
            ctx.track_async(2 * lp, 2 * lp + 1, FB_IN0 + lp, V_OUT0 + lp, NFEAT)
            ctx.featbuf_download(V_OUT0 + lp, NFEAT)
        ms_pcie = (time.perf_counter() - t) / reps * 1e3
        # pipelined ingest: frames already sit in pinned host memory (as a decoder would leave them), uploads run on the
        # copy stream and overlap the previous pair's kernels; records go to a device table read back every 16 pairs
        NPIN = min(PL, 8)
        pins = []
        for lp in range(NPIN):
            a, b = ctx.pinned_array((HEIGHT, WIDTH)), ctx.pinned_array((HEIGHT, WIDTH))
            a[:], b[:] = frames[pair_index(0, lp)]
            pins.append((a, b))
        # records: a device table of 2 x 16 rows; the half a window filled goes to pinned host memory with klt_featbuf_download_async at the
        # window's end and is waited for at the NEXT window's end -- a synchronous download there makes the host wait for every queued step
        # and the link idles 0.3-0.9 ms per window meanwhile (tools/trace_copies.py)
        TAB, NT = FB_MISC + 1, 16
        HALVES = (TAB + 1 + 2 * NT, TAB + 2 + 2 * NT)
        ctx.featbuf_alloc(TAB, 2 * NT * NFEAT)
        for k in range(2 * NT):
            ctx.featbuf_view(TAB + 1 + k, TAB, k * NFEAT, NFEAT)
        for hlf in range(2):
            ctx.featbuf_view(HALVES[hlf], TAB, hlf * NT * NFEAT, NT * NFEAT)
        from pyfeaturetrack_amd.backend import FEAT_DTYPE
        host_tab = [ctx.pinned_array((NT * NFEAT,), FEAT_DTYPE) for _ in range(2)]
        npipe = 16 * NT

        def send(i):                         # the two frames of pair i leave on the two copy streams
            lp = i % NPIN
            ctx.upload_async(2 * lp, pins[lp][0])
            ctx.upload_async(2 * lp + 1, pins[lp][1])

        def pipelined_step(i):
            # the NEXT pair's frames are sent before this pair's kernels are enqueued: the link works on pair i + 1 while the GPU works on
            # pair i (four pairs of slots in rotation; a slot's raw buffers alternate, so the copy never waits for the build before last)
            lp = i % NPIN
            send(i + 1)
            ctx.build_pyramids_batch([2 * lp, 2 * lp + 1])
            ctx.track_async(2 * lp, 2 * lp + 1, FB_IN0 + lp, TAB + 1 + i % (2 * NT), NFEAT)
            if i % NT != NT - 1:
                return None
            win = i // NT
            ctx.download_wait()                                   # the PREVIOUS window's records (long there)
            got = host_tab[(win - 1) % 2] if win > 0 else None     # (valid until the window after next overwrites it)
            ctx.featbuf_download_async(HALVES[win % 2], host_tab[win % 2])
            return got

        send(0)
        for i in range(NT):                 # warm-up: the alternate raw buffers are allocated on first use
            table = pipelined_step(i)
        ctx.sync()
        t = time.perf_counter()
        for i in range(NT, NT + npipe):
            got = pipelined_step(i)
            table = got if got is not None else table
        ctx.download_wait()
        table = host_tab[((NT + npipe - 1) // NT) % 2].copy()     # the last window's records
        ctx.sync()
        ms_pipe = (time.perf_counter() - t) / npipe * 1e3
        last_lp = (NT + npipe - 1) % NPIN
        assert np.array_equal(table[-NFEAT:]["x"], outs[pair_index(0, last_lp)]["x"]), "pipelined ingest changed the result"
        extra = {"region_ms_per_step": reg,
                 "overlapped_ms_per_pair": ms_per_pair,
                 "cache_resident_ms_per_pair": ms_hot, "cache_resident_features_per_s": NFEAT / (ms_hot * 1e-3),
                 "pcie_pipelined_ms_per_pair": ms_pipe, "pcie_pipelined_features_per_s": NFEAT / (ms_pipe * 1e-3),
                 "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3, "latency_ms_per_pair_synchronised": ms_latency,
                 "single_stream_ms_per_pair": ms_single, "single_stream_features_per_s": NFEAT / (ms_single * 1e-3),
                 "single_stream_runs_ms": singles,
                 "ms_per_select_5000": ms_select,
                 "pcie_inclusive_ms_per_pair": ms_pcie, "pcie_inclusive_features_per_s": NFEAT / (ms_pcie * 1e-3),
                 "note": "ms_per_frame_pair = single_stream_ms_per_pair: one pair at a time on ONE stream, rotating through the resident "
                         "pairs, no overlap with other pairs (ms_per_step / pairs_per_step = overlapped_ms_per_pair is the inverse "
                         "throughput with pairs_in_flight pairs overlapping).  cache_resident = round 2's headline arrangement: every "
                         "context rebuilds the same four frame slots, which then never leave the 256 MB Infinity Cache.  pcie_inclusive "
                         "= H2D of two u8 frames from pageable host memory + pyramids + track + D2H of the records, synchronised per "
                         "pair; pcie_pipelined = the same bytes with klt_upload_u8_async from pinned memory on two copy streams, the next "
                         "pair sent before this pair's kernels are enqueued, and the records read back every 16 pairs without draining the "
                         "queue (klt_featbuf_download_async)"}
        if tree:
            extra["tracker_tree_sums"] = tree
        link = link_rates()
        extra["pcie_pipelined_GBps"] = 2 * WIDTH * HEIGHT / (ms_pipe * 1e-3) / 1e9
        if link:
            extra["pcie_link"] = link
            extra["pcie_pipelined_frac_of_link"] = extra["pcie_pipelined_GBps"] / link["1080p"]
            extra["pcie_pipelined_frac_of_link_next_to_a_kernel"] = min(1.0, extra["pcie_pipelined_GBps"] / link["1080p_next_to_a_kernel"])

    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline and ko:
        a0, a1 = frames[0][0].astype(np.float32), frames[0][1].astype(np.float32)
        cpu = cpu_baseline_of(ko, lambda: ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), lists[0].copy()), NFEAT,
                              "pyramids of both frames + track 5000 features of ONE pair of cfg-2 (1920x1080, seed %d)" % seeds[0],
                              reference_python_survey={"ms_per_pair": 603.0, "features_per_s": 8300.0, "where": "survey container, 1 thread"})
        if cpu:
            cpu["ms_per_pair"] = cpu["ms_per_step"]

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = base_line(world * NP * NFEAT * args.steps / elapsed, world, args.steps, args.warmup, ms_per_step,
                         ms_single if ms_single is not None else ms_per_pair,
                         ("THROUGHPUT over %d independent pairs in flight (%d contexts x %d pairs per launch); the literal single-pair figure is "
                          "`single_pair`.  " % (nctx * B, nctx, B) if nctx * B > 1 else "ONE pair at a time on one stream.  ") +
                         "cfg-2: %d DISTINCT 1920x1080 synthetic pairs resident per GPU (seeds %d..%d; own frame slots, pyramids and "
                         "feature lists: %.1f GB), 5000 features each, 7x7 window, 3 pyramid levels (subsampling 4), translation only; "
                         "a step = one pass of pyramid build + tracking over all of them (%d KLTTrackFeatures-equivalents); inputs "
                         "resident in HBM, no frame or pyramid is touched twice within a step"
                         % (NP, seeds[0], seeds[-1], NP * 2 * (WIDTH * HEIGHT + 4 * 3 * sum(level_pixels(p, WIDTH, HEIGHT))) / 1e9, NP),
                         extra_cfg={
                             "pipelining": (("none (one HIP stream)" if nctx == 1 else
                                             "groups of pairs go round-robin to %d contexts, one HIP stream each, no ordering between them "
                                             "(pairs are independent)" % nctx) +
                                            ("; every pair has its own launches" if B == 1 else
                                             "; the %d pairs of a group share every launch of their context: one batched pyramid "
                                             "build for their %d frames, one tracker launch for their %d feature lists -- every pair still "
                                             "gets the full work of one KLTTrackFeatures call" % (B, 2 * B, B))),
                             "pairs_in_flight": nctx * B, "contexts": nctx, "pairs_per_launch": B, "resident_pairs": NP,
                             "features_per_pair": NFEAT, "pairs_per_step": NP * world, "ms_per_pair": ms_per_pair, "tracked": tracked,
                             "recovered_shift_px": shift, "imposed_shift_px": list(synth.DEFAULT_SHIFT),
                             "rccl_ranks": rccl.get("rccl_ranks", 0),
                             "parallelism": "%d pairs per GPU" % NP + (", one RCCL all-gather (libkltgpu side stream) of each context's [%d pairs x "
                                                                      "5000] record table per step" % PL if distributed else "")})
        line.update(parity)
        # BASELINE cfg-2 read literally -- "single 1920x1080 pair": one pair at a time on ONE stream, nothing overlapping it
        if ms_single is not None:
            step_b = roofline["step_algorithmic_bytes_formula"] if roofline else None
            line["single_pair"] = {"ms": ms_single, "features_per_s": NFEAT / (ms_single * 1e-3),
                                   "step_frac": (step_b / (ms_single * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_b else None,
                                   "note": "pyramids of both frames + tracker of ONE pair per build / tracker call on one HIP stream, rotating "
                                           "through the resident pairs (median of 5 runs of %d pairs); `value` is the throughput with %d "
                                           "independent pairs in flight" % (4 * PL, nctx * B)}
        line["roofline"], line["cpu_baseline"], line["extra"] = roofline, cpu, extra
        if rccl and isinstance(extra, dict):
            extra["rccl_validation"] = rccl
    for cx in ctxs:
        cx.close()
    # The figures a caller of the API / of a sequence loop sees are taken in a process that holds nothing else: the headline's contexts
    # (twelve HIP streams, 4.2 GB of slots) are closed first -- with them open the same calls read 5-15 % slower (more queues than the
    # hardware schedules side by side).  One-GPU secondary figures: with N > 1 the other ranks are done by now and must not be kept
    # waiting for rank 0's extras.
    if line is not None and isinstance(extra, dict) and not args.no_extras and not distributed:
        if not args.no_api:
            extra.update(api_figures_in_a_child_process())
            # ... and once more with the child confined to eight CPUs of the GPU's NUMA node, as a deployment would pin it: the host part of a
            # call (frame comparison on the pool's lanes, column moves) stops migrating between the box's 256 CPUs -- the same figures with a
            # third of the run-to-run spread.  The unpinned ones above stay the headline of the API.
            cpus = gpu_numa_cpus(ranks.local_rank)
            if cpus and "api_error" not in extra:
                pinned = api_figures_in_a_child_process(cpus=cpus)
                extra["api_pinned_to_the_gpus_numa_node"] = dict({k: v for k, v in pinned.items() if k.startswith("api_ms") or k == "api_error"},
                                                                 cpus=cpus)
        if not args.no_sequences:
            link = extra.get("pcie_link") or {}
            extra["sequence_from_host"] = {"1080p": sequence_from_host(ranks.local_rank, 1920, 1080, 5000, 256, link.get("1080p")),
                                           "4k": sequence_from_host(ranks.local_rank, 3840, 2160, 20000, 128, link.get("4k")),
                                           "note": sequence_from_host.__doc__.split("  Secondary")[0].replace("\n    ", " ")}
    sweep_failed = []
    if line is not None and isinstance(extra, dict) and not args.no_extras and not distributed and not args.no_config_sweep:
        # the other four BASELINE configs, each in a child process of its own (benchlib/sweep.py): compact records in extra.configs
        from .sweep import config_sweep
        extra["configs"], sweep_failed = config_sweep(["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    if line is None:
        Ranks.fail_on_validation(rccl)                  # (the other ranks leave non-zero as well: the launcher then reports the run as failed)
    if line is not None:
        emit(json_fd, line)
        fail_on_parity(parity)
        Ranks.fail_on_validation(rccl)
        if sweep_failed:
            raise SystemExit("config sweep: %s failed: %s" % (", ".join(sweep_failed), "; ".join(extra["configs"][c]["error"] for c in sweep_failed)))


SEQ_ARRANGEMENTS = {"two_copy_streams": 2, "one_copy_stream": 1}      # name -> KLT_OPT_COPY_STREAMS
SEQ_KEPT = "one_copy_stream"                                           # what KLTTrackSequence uses (profiles/README.md, round 6)


def sequence_from_host(device, w, h, n, nframes=256, link_gbps=None, alternations=5, only=None):
    """What a video pipeline pays per frame when the frames come from the host: sequential mode, ONE new u8 frame per
    step from pinned host memory (klt_upload_u8_async on the copy streams, overlapping the previous frame's kernels), pyramid of the new
    frame + score preparation on the build stream, track + replacement of the lost features on the main stream, the next tracker enqueued
    ahead of the host's look -- the loop of `--config cfg5` with an upload per frame -- and the records written into a device table of 16
    rows that is downloaded every 16 frames.  16 distinct frames of the periodic texture sit in pinned memory and are visited up and down
    (0, 1, ... 15, 14, ... 0, ...), so consecutive frames always differ by one step of (3.3, -2.1) pixels.  Secondary figure, never `value`.
    The loop is run in two arrangements alternately, `alternations` times each after a discarded first call: consecutive
    uploads alternating between two copy streams (rounds 3-5, still the default for pairs) and all on ONE copy stream; `ms_per_frame` is the
    median of the arrangement the product's loop (KLTTrackSequence) uses, SEQ_KEPT.  (Sending frame k + 3 earlier -- right behind
    klt_select_begin_async -- was the other half of the experiment: no gain on one stream, a loss on two; profiles/README.md.)"""
    tc = cfg2_context()
    tc.max_residue = 10.0
    ctx = Context(device)
    ctx.configure(tc)
    try:
        NPIN, NT = 16, 16
        phases = synth.sequence_phases(w, h, 4, workers=usable_cores(10))
        pins = []
        for f in synth.periodic_sequence(w, h, 4, NPIN, phases=phases):
            a = ctx.pinned_array((h, w))
            a[:] = f
            pins.append(a)
        order = list(range(NPIN)) + list(range(NPIN - 2, 0, -1))              # 0..15..1: period 30
        S = [0, 1, 2]
        TAB, HALF = 100, (200, 201)                                           # 2 x 16 rows: one half fills while the other is read back
        ctx.featbuf_alloc(TAB, 2 * NT * n)
        for k in range(2 * NT):
            ctx.featbuf_view(TAB + 1 + k, TAB, k * n, n)
        for i in range(2):
            ctx.featbuf_view(HALF[i], TAB, i * NT * n, NT * n)
        row = lambda k: TAB + 1 + k % (2 * NT)                                # noqa: E731
        from pyfeaturetrack_amd.backend import FEAT_DTYPE
        host_tab = [ctx.pinned_array((NT * n,), FEAT_DTYPE) for _ in range(2)]
        ctx.set_option(15, 1)                                                 # KLT_OPT_BUILD_STREAM

        def send(k):                         # frame k leaves for its slot (the copy overlaps whatever the GPU is doing)
            ctx.upload_async(S[k % 3], pins[order[k % len(order)]])

        def stage(k):
            ctx.build_pyramids(S[k % 3], sync=False)
            ctx.select_prepare(S[k % 3])

        def track(k):
            ctx.track_async(S[(k - 1) % 3], S[k % 3], row(k - 1), row(k), n)

        def run(count):
            live = None
            send(0)
            ctx.build_pyramids(S[0], sync=False)
            ctx.select_async(S[0], 1, True, row(0), n)
            send(1)
            send(2)
            stage(1)
            track(1)
            send(3)
            for k in range(1, count):
                ctx.select_begin(S[k % 3], 2, True, row(k), n)
                stage(k + 1)
                track(k + 1)
                if ctx.select_finish():
                    track(k + 1)
                # frame k + 3 goes into the slot of frame k, whose pyramids only the tracker just enqueued (k -> k + 1) still reads: the copy fills
                # the slot's other raw buffer, two frame times before its build needs it (a frame sent one step ahead is not there
                # in time: 155 us of copy + the build = the whole frame time at 4K).  After the look: a repeated tracker needs slot k valid.
                send(k + 3)
                if k % NT == NT - 1:
                    # the half holding rows k-15 .. k is complete once frame k's selection is; the tracker of k+1 already writes into the
                    # other half.  The copy is enqueued in stream order and waited for one window later: the host never drains the queue.
                    ctx.download_wait()
                    ctx.featbuf_download_async(HALF[(k // NT) % 2], host_tab[(k // NT) % 2])
                    live = host_tab[(k // NT) % 2]
            ctx.download_wait()
            ctx.sync()
            return None if live is None else live.copy()

        def timed(name):
            ctx.sync()
            ctx.set_option(20, SEQ_ARRANGEMENTS[name])                        # KLT_OPT_COPY_STREAMS
            t = time.perf_counter()
            tab = run(nframes)
            return (time.perf_counter() - t) / (nframes - 1) * 1e3, tab

        run(2 * NT)                                                           # sizes every buffer
        timed(only or SEQ_KEPT)                                               # (the first full call of a process reads 10-20 % high whatever its order: discarded)
        names = [only] if only else list(SEQ_ARRANGEMENTS)            # (`only`: a profiler's pass over one arrangement)
        runs = {name: [] for name in names}
        tables = {}
        for _ in range(max(1, alternations)):
            for name in names:
                ms_one, tables[name] = timed(name)
                runs[name].append(ms_one)
        ctx.set_option(20, 2)
        kept = only or SEQ_KEPT
        assert all(np.array_equal(tables[kept], tab) for tab in tables.values()), "the two arrangements of the loop gave different records"
        table = tables[kept]
        med = {name: statistics.median(v) for name, v in runs.items()}
        ms = med[kept]
        alive = int((table.reshape(NT, n)[NT - 2]["val"] >= 0).sum())
        gbps = w * h / (ms * 1e-3) / 1e9
        out = {"ms_per_frame": ms, "features_per_s": n / (ms * 1e-3), "frames": nframes, "ingest_GBps": gbps,
               "alive_after_replacement": alive, "frame": "%dx%d" % (w, h), "features": n,
               "arrangement": kept,
               "arrangements_ms_per_frame": {name: {"median": med[name], "runs": runs[name]} for name in runs}}
        if "two_copy_streams" in med and kept != "two_copy_streams":
            out["kept_over_two_copy_streams"] = med[kept] / med["two_copy_streams"]
        if link_gbps:
            out["link_GBps"] = link_gbps
            out["ingest_frac_of_link"] = gbps / link_gbps
        return out
    finally:
        ctx.close()


def link_rates():
    """profiles/r04_h2d_probe.json (tools/h2d_probe.cpp on the builder's GPU box): what pinned host-to-device copies of one frame sustain"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r04_h2d_probe.json")))
        return {"1080p": d["h2d_1080p_2.07MB"]["two_streams_GBps"], "4k": d["h2d_4k_8.29MB"]["two_streams_GBps"],
                "1080p_next_to_a_kernel": d["h2d_1080p_2.07MB"]["two_streams_next_to_a_kernel_GBps"],
                "4k_next_to_a_kernel": d["h2d_4k_8.29MB"]["two_streams_next_to_a_kernel_GBps"],
                "source": "profiles/r04_h2d_probe.json (tools/h2d_probe.cpp, builder gpurun): pinned H2D on two copy streams, back to back, "
                          "on an idle GPU / next to a running compute kernel"}
    except (OSError, KeyError, ValueError):
        return {}


def gpu_numa_cpus(device=0, count=8):
    """the first `count` CPUs of the NUMA node the GPU hangs off (sysfs), or None"""
    try:
        import glob
        nodes = sorted(glob.glob("/sys/class/drm/card*/device/numa_node"))
        node = int(open(nodes[min(device, len(nodes) - 1)]).read())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed][:count]
        return cpus if len(cpus) >= 4 else None
    except (OSError, ValueError, IndexError):
        return None


def api_figures_in_a_child_process(timeout=600, cpus=None):
    """`python -m benchlib.api_figures` as a child process: what a caller of the reference-shaped API sees in a process of its OWN -- this
    one has run the headline, the parity checks (32 OpenMP threads of the oracle) and a dozen probes, and the same calls read 5-30 % slower
    inside it than in a fresh interpreter (KLTTrackSequence 0.166-0.214 against 0.147 ms per 1080p frame).  The child opens the GPU itself;
    this process only waits for it.  `cpus`: the child confines itself to these CPUs before it touches the GPU (sched_setaffinity)."""
    import subprocess
    try:
        env = dict(os.environ)
        if cpus:                                            # (the child confines itself first thing: no code between fork and exec here)
            env["KLT_API_FIGURES_CPUS"] = ",".join(str(c) for c in cpus)
        r = subprocess.run([sys.executable, "-m", "benchlib.api_figures"], cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"api_error": "benchlib.api_figures exited with %d: %s" % (r.returncode, r.stderr[-400:])}
        out = json.loads(lines[-1])
        out["api_measured_in"] = "a child process of its own (python -m benchlib.api_figures), started after this process had closed its contexts"
        return out
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"api_error": "%s: %s" % (type(e).__name__, e)}
