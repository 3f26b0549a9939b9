"""bench.py --config cfg1: img0 -> img1, 100 features (the reference's own CPU-runnable case)."""
from .common import *            # noqa: F401,F403 -- the shared helpers, constants and the modules they import (np, os, time, ...)


def run_cfg1(args, json_fd):
    """BASELINE cfg-1: img0.pgm -> img1.pgm, 100 features, default context (7x7, 2 levels / ss 4), max_residue 10."""
    from tests.conftest import read_pgm
    g = os.path.join(ROOT, "tests", "golden")
    i0, i1 = read_pgm(os.path.join(g, "img0.pgm")), read_pgm(os.path.join(g, "img1.pgm"))
    n = 100
    tc = KLT_TrackingContext()
    tc.max_residue = 10.0
    p = params_from_tc(tc)
    ctx = Context(0)
    ctx.configure(tc)
    ctx.upload(0, i0)
    ctx.upload(1, i1)
    ctx.build_pyramids_batch([0, 1], sync=True)
    ctx.select(0, n, use_pyramid=True)                 # first call allocates the selection scratch
    t = time.perf_counter()
    fl, _ = ctx.select(0, n, use_pyramid=True)
    ms_select = (time.perf_counter() - t) * 1e3
    ctx.featbuf_upload(0, fl)

    def step():
        ctx.build_pyramids_batch([0, 1])
        ctx.track_async(0, 1, 0, 1, n)

    def region():
        for _ in range(args.steps):
            step()

    for _ in range(max(1, args.warmup)):
        step()
    el, regions, enq = timed_regions(OneGpu([ctx]), region, args.repeats)
    out = ctx.featbuf_download(1, n)
    ko = load_oracle()
    checks = []
    if ko:
        ofl = ko.select_good_features(p, i0.astype(np.float32), n)
        same_sel = bool(np.array_equal(fl["x"], ofl["x"]) and np.array_equal(fl["y"], ofl["y"]) and np.array_equal(fl["val"], ofl["val"]))
        checks.append(("selection of 100 on img0", same_sel, 0.0))
        same, dx = records_equal(out, oracle_track(ko, p, i0, i1, fl))
        checks.append(("100 features tracked img0 -> img1", same, dx))
    par = parity_summary(checks, "selected list and the tracked records of the last timed step")
    nst = min(args.steps, 50)

    def plain():
        for _ in range(nst):
            step()

    plain()
    ctx.sync()
    ctx.track_stats_reset()
    paired = timed_pass(ctx, plain, 1)
    st = ctx.track_stats()
    sane_iterations(st, nst * int((fl["val"] >= 0).sum()), p.nPyramidLevels, "cfg-1")
    stamped = timed_pass(ctx, plain, 2)
    ms_step = el / args.steps * 1e3
    roof = roofline_of(kernel_table(stamped, paired, nst, {"track": track_bytes(p, st, nst * n)}), nst, ms_step)
    cpu = None
    if ko and not args.no_cpu_baseline:
        a0, a1 = i0.astype(np.float32), i1.astype(np.float32)
        cpu = cpu_baseline_of(ko, lambda: ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), fl.copy()), n,
                              "pyramids of img0 and img1 + track 100 features (cfg-1)", budget_s=5.0)
    ctx.close()
    line = base_line(n * args.steps / el, 1, args.steps, args.warmup, ms_step, ms_step,
                     "cfg-1: img0.pgm -> img1.pgm (320x240), 100 features, 7x7, 2 levels (ss 4), max_residue 10; per step: pyramids of "
                     "both frames + track",
                     extra_cfg={"tracked": int((out["val"] >= 0).sum()), "ms_select_100": ms_select})
    line.update(par)
    line["roofline"], line["cpu_baseline"] = roof, cpu
    line["extra"] = {"region_ms_per_step": region_stats(regions, args.steps, el), "host_enqueue_ms_per_step": enq / args.steps * 1e3}
    emit(json_fd, line)
    fail_on_parity(par)

