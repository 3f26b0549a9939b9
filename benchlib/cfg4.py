"""bench.py --config cfg4: the 256-pair batch of 1280x720 frames, sharded over the ranks with one gather per step."""
from .common import *            # noqa: F401,F403 -- the shared helpers, constants and the modules they import (np, os, time, ...)


def run_cfg4(args, json_fd):
    """BASELINE cfg-4: 256 independent 1280x720 pairs (seeds 0..255), 2000 features each, 7x7, 3 levels / ss 4, sharded
    contiguously over the ranks (32 per GPU at N = 8; shards may differ by one pair), frames resident in HBM.  Per step every rank
    builds the pyramids of its whole shard (frames share launches through blockIdx.z), tracks it with ONE launch into a device-side
    [pairs x features] table and the table is gathered to rank 0 with one RCCL gather (a count per rank).  Total work is fixed:
    strong scaling."""
    ranks = Ranks(args)
    total, w, h, nf = args.pairs, 1280, 720, 2000
    mine = parallel.shard_range(total, ranks.world, ranks.rank)
    pairs = len(mine)
    tc = cfg2_context()
    p = params_from_tc(tc)
    ctx = Context(ranks.local_rank)
    ctx.set_params(p)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, usable_cores(16) // max(1, min(ranks.world, 8)))) as ex:
        frames = list(ex.map(lambda i: synth.synth_pair(w, h, seed=i), mine))
    for k, (f0, f1) in enumerate(frames):
        ctx.upload(2 * k, f0)
        ctx.upload(2 * k + 1, f1)
    slots = list(range(2 * pairs))
    T_IN, T_OUT, T_ALL, V_IN, V_OUT = 0, 1, 2, 1000, 1000 + max(pairs, 1)
    if pairs:
        ctx.build_pyramids_batch(slots, sync=True)
        ctx.featbuf_alloc(T_IN, pairs * nf)
        ctx.featbuf_alloc(T_OUT, pairs * nf)
    for k in range(pairs):
        ctx.featbuf_view(V_IN + k, T_IN, k * nf, nf)
        ctx.featbuf_view(V_OUT + k, T_OUT, k * nf, nf)
        ctx.select_async(2 * k, 1, True, V_IN + k, nf)
    ctx.sync()
    table = [(2 * k, 2 * k + 1, V_IN + k, V_OUT + k) for k in range(pairs)]
    ranks.attach([ctx])
    gather = parallel.ShardGather(ctx, T_OUT, T_ALL, total, nf, root=0) if ranks.distributed else None

    def step():
        if pairs:
            ctx.build_pyramids_batch(slots)
        if gather and pairs:
            ctx.comm_fence_featbuf(T_OUT)          # the gather of the previous step has read the table
        if pairs:
            ctx.track_batch_async(table, nf)
        if gather:
            gather.gather_async()

    def region():
        for _ in range(args.steps):
            step()

    for _ in range(max(1, args.warmup)):
        step()
    el, regions, enq = timed_regions(ranks, region, args.repeats)
    rccl = ranks.validation(args.steps)
    # what was timed, against the oracle: the first and the last pair of rank 0's shard
    out = ctx.featbuf_download(T_OUT, pairs * nf).reshape(pairs, nf) if pairs else np.zeros((0, nf), parallel.FEAT_DTYPE)
    ko = load_oracle() if ranks.rank == 0 else None
    par = {}
    fl_in = ctx.featbuf_download(T_IN, pairs * nf).reshape(pairs, nf) if pairs else None
    if ranks.rank == 0 and pairs:
        checks = []
        for k in range(pairs) if ko else []:
            same, dx = records_equal(out[k], oracle_track(ko, p, frames[k][0], frames[k][1], fl_in[k], threads=usable_cores()))
            checks.append(("pair %d" % mine[k], same, dx))
        par = parity_summary(checks, "tracked records of ALL %d pairs of rank 0's shard, last timed step" % pairs)
    gathered_ok = None
    if gather:
        full = gather.result()
        if ranks.rank == 0:
            gathered_ok = bool(full.shape == (total, nf) and np.array_equal(full[:pairs], out))
            if ko and ranks.world > 1 and par.get("parity_checked") is not None:
                # what another rank contributed, against the oracle: the last pair of the batch (the last rank's shard), selected and
                # tracked on the CPU from its seed
                g0, g1 = synth.synth_pair(w, h, seed=total - 1)
                ko.set_threads(usable_cores())
                osel = ko.select_good_features(p, g0.astype(np.float32), nf)
                ko.set_threads(1)
                same, dx = records_equal(full[total - 1], oracle_track(ko, p, g0, g1, osel, threads=usable_cores()))
                checks.append(("pair %d as gathered from rank %d" % (total - 1, ranks.world - 1), same, dx))
                par = parity_summary(checks, "tracked records of all %d pairs of rank 0's shard and of the batch's last pair as gathered" % pairs)
    roof = cpu = None
    if ranks.rank == 0 and pairs:
        nst = min(args.steps, 10)

        def plain_steps():
            for _ in range(nst):
                ctx.build_pyramids_batch(slots)
                ctx.track_batch_async(table, nf)

        plain_steps()                                   # (the parity check left the GPU idle)
        ctx.sync()
        ctx.track_stats_reset()
        paired = timed_pass(ctx, plain_steps, 1)
        st = ctx.track_stats()
        sane_iterations(st, nst * pairs * nf, p.nPyramidLevels, "cfg-4")
        stamped = timed_pass(ctx, plain_steps, 2)
        kt = kernel_table(stamped, paired, nst, {"track": track_bytes(p, st, nst * pairs * nf)})
        # the line's step is the whole batch on all ranks: rank 0's kernels describe its own shard
        ms_step = el / args.steps * 1e3
        roof = roofline_of(kt, nst, ms_step)
        rank0_bytes = roof["step_algorithmic_bytes"]
        step_bytes = rank0_bytes * total / pairs
        roof.update({"peak": HBM_PEAK_GBS * ranks.world, "step_algorithmic_bytes": step_bytes, "rank0_shard_algorithmic_bytes": rank0_bytes,
                     "step_frac": step_bytes / (ms_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * ranks.world),
                     "kernels_note": "rank 0's shard (%d of %d pairs), per-GPU peak %g GB/s" % (pairs, total, HBM_PEAK_GBS)})
        for k in roof["kernels"].values():
            k["frac"] = k["GBps"] / HBM_PEAK_GBS
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["kernel_peak"] = HBM_PEAK_GBS
        if ko and not ranks.distributed and not args.no_cpu_baseline:
            a0, a1 = frames[0][0].astype(np.float32), frames[0][1].astype(np.float32)
            cpu = cpu_baseline_of(ko, lambda: ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), fl_in[0].copy()), nf,
                                  "pyramids of both frames + track 2000 features of ONE 1280x720 pair of cfg-4 (seed %d)" % mine[0])
    ctx.close()
    if ranks.rank == 0:
        ms_step = el / args.steps * 1e3
        tracked = int(np.count_nonzero(out["val"] >= 0))
        line = base_line(total * nf * args.steps / el, ranks.world, args.steps, args.warmup, ms_step, ms_step / total,
                         "cfg-4: %d independent 1280x720 pairs per step (%d on rank 0), 2000 features each, 7x7, 3 levels "
                         "(subsampling 4); per rank: batched pyramid build + one tracker launch + one RCCL gather of the "
                         "[pairs x 2000] record table to rank 0" % (total, pairs), scaling="strong",
                         extra_cfg={"pairs_per_step": total, "pairs_per_rank": [len(parallel.shard_range(total, ranks.world, r)) for r in range(ranks.world)],
                                    "tracked_rank0": tracked,
                                    "rccl_ranks": rccl.get("rccl_ranks", 0), "gathered_table_ok": gathered_ok,
                                    "parallelism": "pairs sharded contiguously (shards differ by at most one pair); no data-path collective, one gather with a count per rank"})
        line.update(par)
        line["roofline"], line["cpu_baseline"] = roof, cpu
        line["extra"] = {"region_ms_per_step": region_stats(regions, args.steps, el), "host_enqueue_ms_per_step": enq / args.steps * 1e3}
        if rccl:
            line["extra"]["rccl_validation"] = rccl
        emit(json_fd, line)
        if gathered_ok is False:
            raise SystemExit("the gathered table differs from the shards")
        fail_on_parity(par)
    Ranks.fail_on_validation(rccl)

