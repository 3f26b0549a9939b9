#!/usr/bin/env python3
"""Headline benchmark: features tracked / s and ms per frame pair (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A *step* is one KLTTrackFeatures-equivalent on one frame pair per rank: build the image / gradx /
grady pyramids of both frames from the u8 frames already resident in HBM, then track every live
feature coarse-to-fine (device-resident feature records in, device-resident records out).
Workload at every N = BASELINE cfg-2 (1920x1080 synthetic pair, 5000 features, 7x7 window,
3 pyramid levels / subsampling 4, translation only); with N > 1 every rank runs its own pair
(seed = rank + 1: weak scaling, the path shards by frame pair with no data-path exchange) and the
16-byte feature records of 128 consecutive steps are collected in a device-side table and gathered to
every rank with one RCCL all-gather on a side stream (event-ordered behind the tracker launch,
overlapped with the next steps' kernels).

Consecutive steps go round-robin to `--inflight` contexts (default 3; one HIP stream each, nothing ordering them): frame pairs
are independent, so the GPU overlaps the kernels of different pairs.  Every step does the full work of one pair; `--inflight 1`
and `extra.single_stream_ms_per_pair` give the one-stream figure.  Before the W warm-up steps 60 ms of untimed steps bring the
GPU to its steady state (`--prewarm-ms`).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel of the step (largest share of device time), timed with HIP
                  events on the context's stream in a second pass over the same K steps (events
                  around every launch would distort the un-instrumented `value`);
  cpu_baseline -- the CPU oracle (oracle/klt_oracle.c, a bit-exact port of the reference's
                  Python/Cython/SciPy path) on the same workload, 1 thread, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pyfeaturetrack_amd import synth                                   # noqa: E402
from pyfeaturetrack_amd.backend import Context                          # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                  # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc                    # noqa: E402

WIDTH, HEIGHT, NFEAT = 1920, 1080, 5000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FB_SEL, FB_OUT0, FB_OUT1 = 0, 1, 2
FB_RING0, FB_RING1, FB_VIEW0 = 3, 4, 100
GATHER_EVERY = 128


def cfg2_context():
    tc = KLT_TrackingContext()
    tc.nPyramidLevels = 3
    tc.subsampling = 4
    tc.KLTUpdateTCBorder()          # border 120
    return tc


def algorithmic_bytes(p, ncols, nrows, stats, nfeat):
    """minimum traffic per pair = 2 * bytes_pyramid + sum over features of bytes_track."""
    ss, L = p.subsampling, p.nPyramidLevels
    n, dims = [], (ncols, nrows)
    for _ in range(L):
        n.append(dims[0] * dims[1])
        dims = (dims[0] // ss, dims[1] // ss)
    pyr = n[0] * (1 + 4) + sum(4 * (n[l - 1] + n[l]) for l in range(1, L)) + sum(12 * v for v in n)
    foot = 12.0 * (p.window_width + 1) * (p.window_height + 1)
    track = foot * (sum(stats["level_visits"][:L]) + sum(stats["iterations"][:L])) + 24.0 * nfeat
    return pyr, track


def usable_cores(cap=32):
    """Cores this process can really run on: scheduler affinity, clipped by the cgroup CPU quota and by `cap`."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(p, f0, f1, fl):
    """Oracle timed on the host: bounded sample of the same workload (about 10-20 s of CPU work)."""
    from oracle import klt_oracle as ko
    a0, a1 = f0.astype(np.float32), f1.astype(np.float32)

    def one_pair():
        P0, P1 = ko.Pyramids(p, a0), ko.Pyramids(p, a1)
        return ko.track_features(p, P0, P1, fl.copy())

    ko.set_threads(1)
    t = time.perf_counter()
    one_pair()
    t1 = time.perf_counter() - t
    reps = int(max(2, min(200, 12.0 / max(t1, 1e-3))))      # about 12 s of single-thread work
    t = time.perf_counter()
    for _ in range(reps):
        one_pair()
    dt = (time.perf_counter() - t) / reps
    # the same port on the host cores this process may actually use (OpenMP over image lines / features;
    # bit-identical results).  Time-bounded: a container with a CPU quota can make many threads slower than one.
    ncores = ko.set_threads(usable_cores())
    t = time.perf_counter()
    reps_all = 0
    while reps_all < 40 and (reps_all < 2 or time.perf_counter() - t < 4.0) and time.perf_counter() - t < 20.0:
        one_pair()
        reps_all += 1
    dt_all = (time.perf_counter() - t) / reps_all
    ko.set_threads(1)
    return {"value": NFEAT / dt, "unit": "features/s", "cores": 1, "kind": "port",
            "ms_per_pair": dt * 1e3,
            "sample": "%d x (pyramids of both 1920x1080 frames + track 5000 features), oracle/klt_oracle.c, 1 thread" % reps,
            "all_cores": {"value": NFEAT / dt_all, "cores": ncores, "ms_per_pair": dt_all * 1e3,
                          "sample": "%d x the same pair, OpenMP over image lines and features" % reps_all}}


def run_cfg4(args, json_fd):
    """BASELINE cfg-4 (not the headline line): a shard of independent 1280x720 pairs, 2000 features each, 7x7,
    3 levels / ss 4.  One batched pyramid build (frames share launches through blockIdx.z) and ONE tracker launch
    per step; shows what the kernels do when the grids are large."""
    pairs, w, h, nf = args.pairs, 1280, 720, 2000
    tc = cfg2_context()
    p = params_from_tc(tc)
    ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    ctx.set_params(p)
    for i in range(pairs):
        f0, f1 = synth.synth_pair(w, h, seed=i)
        ctx.upload(2 * i, f0)
        ctx.upload(2 * i + 1, f1)
    slots = list(range(2 * pairs))
    ctx.build_pyramids_batch(slots, sync=True)
    for i in range(pairs):
        ctx.select_async(2 * i, 1, True, 2 * i, nf)
    ctx.sync()
    table = [(2 * i, 2 * i + 1, 2 * i, 2 * i + 1) for i in range(pairs)]

    def step():
        ctx.build_pyramids_batch(slots)
        ctx.track_batch_async(table, nf)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    ctx.track_stats_reset()
    ctx.timing_enable(True)
    for _ in range(args.steps):
        step()
    kernels = ctx.timing_read()
    ctx.timing_enable(False)
    st = ctx.track_stats()
    st = {k: ([x / (args.steps * pairs) for x in v] if isinstance(v, list) else v / (args.steps * pairs)) for k, v in st.items()}
    pyr_bytes, track_bytes = algorithmic_bytes(p, w, h, st, nf)
    step_bytes = pairs * (2 * pyr_bytes + track_bytes)
    tracked = sum(int(np.count_nonzero(ctx.featbuf_download(2 * i + 1, nf)["val"] >= 0)) for i in range(pairs))
    line = {"metric": "features tracked/sec", "value": pairs * nf * args.steps / elapsed, "unit": "features/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_frame_pair": elapsed / args.steps / pairs * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (convolutions accumulate in f64)", "data": "synthetic",
            "config": {"workload": "cfg-4 shard: %d independent 1280x720 pairs per step, 2000 features each, 7x7, 3 levels "
                                   "(subsampling 4); batched pyramid build + one tracker launch" % pairs,
                       "pairs_per_step": pairs, "tracked": tracked},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "step_algorithmic_bytes": step_bytes,
                         "achieved": step_bytes / elapsed * args.steps / 1e9,
                         "frac": step_bytes / elapsed * args.steps / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernels": {k["name"]: {"us_per_launch": 1e3 * k["total_ms"] / k["launches"],
                                                 "launches_per_step": k["launches"] / args.steps} for k in kernels}},
            "cpu_baseline": None}
    ctx.close()
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def _emit(json_fd, line):
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def _base_line(value, steps, warmup, ms_step, workload, extra_cfg=None, pairs=1):
    cfg = {"workload": workload}
    cfg.update(extra_cfg or {})
    return {"metric": "features tracked/sec", "value": value, "unit": "features/s", "n_gpus": 1, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_step, "ms_per_frame_pair": ms_step / pairs, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (convolutions accumulate in f64)", "data": "synthetic",
            "config": cfg, "roofline": None, "cpu_baseline": None}


def run_cfg1(args, json_fd):
    """BASELINE cfg-1: img0.pgm -> img1.pgm, 100 features, default context (7x7, 2 levels / ss 4), max_residue 10."""
    from tests.conftest import read_pgm
    g = os.path.join(ROOT, "tests", "golden")
    i0, i1 = read_pgm(os.path.join(g, "img0.pgm")), read_pgm(os.path.join(g, "img1.pgm"))
    tc = KLT_TrackingContext()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)
    ctx.upload(0, i0)
    ctx.upload(1, i1)
    ctx.build_pyramids_batch([0, 1], sync=True)
    ctx.select(0, 100, use_pyramid=True)                 # first call allocates the selection scratch
    t = time.perf_counter()
    fl, _ = ctx.select(0, 100, use_pyramid=True)
    ms_select = (time.perf_counter() - t) * 1e3
    ctx.featbuf_upload(0, fl)

    def step():
        ctx.build_pyramids_batch([0, 1])
        ctx.track_async(0, 1, 0, 1, 100)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    el = time.perf_counter() - t0
    out = ctx.featbuf_download(1, 100)
    ctx.close()
    _emit(json_fd, _base_line(100 * args.steps / el, args.steps, args.warmup, el / args.steps * 1e3,
                              "cfg-1: img0.pgm -> img1.pgm (320x240), 100 features, 7x7, 2 levels (ss 4), max_residue 10",
                              {"tracked": int((out["val"] >= 0).sum()), "ms_select_100": ms_select}))


def run_cfg3(args, json_fd):
    """BASELINE cfg-3: 1920x1080, 15x15 window, 4 levels / ss 2 (border 108), 5000 features, affine consistency check
    (mode 2, 15x15 affine window) -- 3-frame sequence; the first call only stores templates, steps time later calls."""
    tc = KLT_TrackingContext()
    tc.window_width = tc.window_height = 15
    tc.nPyramidLevels, tc.subsampling = 4, 2
    tc.KLTUpdateTCBorder()
    tc.affineConsistencyCheck = 2
    n = 5000
    ctx = Context(0)
    ctx.configure(tc)
    base = synth.synth_base(WIDTH, HEIGHT, 1)
    frames = [synth.synth_frame(WIDTH, HEIGHT, 1, k, shift=(1.1, -0.7), base=base) for k in range(3)]
    for k, f in enumerate(frames):
        ctx.upload(k, f)
    ctx.build_pyramids_batch([0, 1, 2], sync=True)
    fl, placed = ctx.select(0, n, use_pyramid=True)
    ctx.affine_alloc(0, n)
    ctx.featbuf_upload(0, fl)
    ctx.track_affine_async(0, 1, 0, 1, n, 0)            # stores the templates
    ctx.sync()
    live1 = int((ctx.featbuf_download(1, n)["val"] >= 0).sum())

    def step():                                          # frame 1 -> frame 2 with the affine check active
        ctx.build_pyramids_batch([1, 2])
        ctx.track_affine_async(1, 2, 1, 2, n, 0)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    el = time.perf_counter() - t0
    out = ctx.featbuf_download(2, n)
    ctx.timing_enable(True)
    for _ in range(min(args.steps, 20)):
        step()
    kern = {k["name"]: round(1e3 * k["total_ms"] / k["launches"], 1) for k in ctx.timing_read()}
    ctx.close()
    _emit(json_fd, _base_line(live1 * args.steps / el, args.steps, args.warmup, el / args.steps * 1e3,
                              "cfg-3: 1920x1080, %d features placed (%d live), 15x15 window, 4 levels (ss 2), affine consistency "
                              "check mode 2 (parity unpinned); per step: pyramids of both frames + translation tracker + affine check"
                              % (placed, live1),
                              {"tracked_after_affine": int((out["val"] >= 0).sum()), "kernel_us": kern}))


def run_cfg5(args, json_fd):
    """BASELINE cfg-5 (single GPU): 3840x2160 sequence, 20000 features, sequential mode, lost features replaced after every
    frame.  Per step: upload is excluded (frames resident), pyramid of the new frame, track, REPLACING_SOME selection."""
    w, h, n = 3840, 2160, 20000
    nframes = 8
    tc = cfg2_context()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)
    base = synth.synth_base(w, h, 4)
    for k in range(nframes):
        ctx.upload(10 + k, synth.synth_frame(w, h, 4, k, base=base))
    ctx.build_pyramids(10)
    fl, placed = ctx.select(10, n, use_pyramid=True)
    ctx.featbuf_upload(0, fl)
    ctx.sync()

    def run_sequence(timed):
        t_sel = 0.0
        for k in range(1, nframes):
            ctx.build_pyramids(10 + k, sync=False)
            ctx.track_async(10 + k - 1, 10 + k, (k - 1) % 2, k % 2, n)
            if timed:
                ctx.sync()
                t = time.perf_counter()
            ctx.select_async(10 + k, 2, True, k % 2, n)       # KLTReplaceLostFeatures on the resident level-0 images
            if timed:
                ctx.sync()
                t_sel += time.perf_counter() - t
        ctx.sync()
        return t_sel

    ctx.featbuf_upload(0, fl)
    run_sequence(False)
    reps = max(1, args.steps // (nframes - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.featbuf_upload(0, fl)
        run_sequence(False)
    el = time.perf_counter() - t0
    ctx.featbuf_upload(0, fl)
    t_sel = run_sequence(True)
    out = ctx.featbuf_download((nframes - 1) % 2, n)
    frames_done = reps * (nframes - 1)
    ctx.close()
    _emit(json_fd, _base_line(n * frames_done / el, frames_done, 0, el / frames_done * 1e3,
                              "cfg-5 (one GPU): 3840x2160 sequence, 20000 features, 7x7, 3 levels (ss 4), sequential mode, lost "
                              "features replaced after every frame; per frame: pyramid of the new frame + track + replacement",
                              {"live_at_end": int((out["val"] >= 0).sum()), "ms_replace_per_frame": t_sel / (nframes - 1) * 1e3}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed hot-path work before the W warm-up steps: the GPU needs ~10 ms of load to reach its steady clocks / "
                         "cache state (a 200-step run right after start-up measures 50 us per pair, the same loop after 50 ms 43 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split-l0", action="store_true", help="KLT_OPT_SPLIT_L0: fork/join pyramid build on two streams")
    ap.add_argument("--config", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the headline line); the others are the remaining BASELINE configs on one GPU, informative")
    ap.add_argument("--pairs", type=int, default=32, help="pairs per step for --config cfg4")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent pairs in flight per GPU: consecutive steps go round-robin to this many contexts (one HIP "
                         "stream each, no events between them), so kernels of different pairs overlap; 1 = a single stream")
    ap.add_argument("--pipeline", action="store_true",
                    help="KLT_OPT_TRACK_STREAM: tracker on a second HIP stream, overlapping the next step's pyramid build "
                         "(measured slower on MI355X for this step size: event cost > overlap gain; DESIGN.md)")
    args = ap.parse_args()

    # stdout must carry exactly one JSON line.  RCCL / the HIP runtime print their own chatter to fd 1 (also at
    # process exit), so fd 1 is pointed at stderr for the whole run and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.config != "cfg2":
        return {"cfg1": run_cfg1, "cfg3": run_cfg3, "cfg4": run_cfg4, "cfg5": run_cfg5}[args.config](args, json_fd)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        print("warning: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
    distributed = world > 1 or os.environ.get("KLT_FORCE_DIST") == "1"    # the env var exercises the RCCL path on one GPU

    torch = dist = None
    if distributed:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    tc = cfg2_context()
    p = params_from_tc(tc)
    f0, f1 = synth.synth_pair(WIDTH, HEIGHT, seed=rank + 1)
    # `--inflight` contexts per GPU, each with its own HIP stream, slots and feature buffers.  Step i runs on context
    # i % inflight: pairs are independent (the path shards by frame pair), so nothing orders the streams against each other
    # and the GPU overlaps the kernels of different pairs -- the drain / ramp between dependent kernels of one pair and the
    # latency-bound tracker are filled with the next pair's convolutions.
    nctx = max(1, args.inflight)
    ctxs, gathers = [], []
    fl = None
    for c in range(nctx):
        cx = Context(local_rank)
        cx.set_params(p)
        # The pair lives in two slot pairs, (0,1) and (2,3), used by alternate steps of a context: with KLT_OPT_TRACK_STREAM
        # the tracker of one step overlaps the pyramid build of the next, which must not overwrite what it reads.
        for s0 in (0, 2):
            cx.upload(s0, f0)
            cx.upload(s0 + 1, f1)
        if args.pipeline:
            cx.set_option(3, 1)
        if args.split_l0:
            cx.set_option(7, 1)
        cx.build_pyramids(0)
        fl_c, placed = cx.select(0, NFEAT, use_pyramid=True)
        assert placed == NFEAT, "only %d of %d features could be placed" % (placed, NFEAT)
        assert fl is None or np.array_equal(fl, fl_c), "contexts selected different features"
        fl = fl_c
        cx.featbuf_upload(FB_SEL, fl)
        cx.featbuf_upload(FB_OUT0, fl)
        cx.featbuf_upload(FB_OUT1, fl)
        # N > 1: the records of GATHER_EVERY consecutive steps of a context land in one device-side [steps x features] table
        # (two tables, used alternately) and each full table is all-gathered with ONE RCCL collective on a side stream --
        # cfg-4's "gather once per shard", and the host cost of a collective is not paid per step.
        if distributed:
            from pyfeaturetrack_amd.parallel import FeatureGather
            for t, ring in enumerate((FB_RING0, FB_RING1)):
                cx.featbuf_alloc(ring, GATHER_EVERY * NFEAT)
                for k in range(GATHER_EVERY):
                    cx.featbuf_view(FB_VIEW0 + t * GATHER_EVERY + k, ring, k * NFEAT, NFEAT)
            gathers.append(FeatureGather(cx, [FB_RING0, FB_RING1], GATHER_EVERY * NFEAT, world, torch, dist))
        ctxs.append(cx)
    ctx = ctxs[0]

    def out_buffer(j):
        """feature buffer that local step j of a context writes"""
        if not distributed:
            return FB_OUT0 if j % 2 == 0 else FB_OUT1
        return FB_VIEW0 + ((j // GATHER_EVERY) % 2) * GATHER_EVERY + j % GATHER_EVERY

    def step(i, last=False):
        c, j = i % nctx, i // nctx                # context, and the step's index among that context's steps
        cx = ctxs[c]
        a = 0 if j % 2 == 0 else 2
        cx.build_pyramids_batch([a, a + 1])       # both frames share every launch
        if not distributed:
            cx.track_async(a, a + 1, FB_SEL, out_buffer(j), NFEAT)
            return
        t, k = (j // GATHER_EVERY) % 2, j % GATHER_EVERY
        ring = FB_RING0 if t == 0 else FB_RING1
        if k == 0:
            gathers[c].wait_free(ring)      # the collective that read this table two rounds ago has finished
        cx.track_async(a, a + 1, FB_SEL, out_buffer(j), NFEAT)
        if k == GATHER_EVERY - 1 or last:
            gathers[c].all_gather(ring)     # RCCL on a side stream, behind this tracker launch (event)

    def fence():
        for cx in ctxs:
            cx.sync()
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # bring the GPU to its steady state first (the same work as a step, into the plain output buffers)
    t_pre = time.perf_counter()
    i_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(32):
            cx = ctxs[i_pre % nctx]
            a = 0 if (i_pre // nctx) % 2 == 0 else 2
            cx.build_pyramids_batch([a, a + 1])
            cx.track_async(a, a + 1, FB_SEL, FB_OUT0 if (i_pre // nctx) % 2 == 0 else FB_OUT1, NFEAT)
            i_pre += 1
        for cx in ctxs:
            cx.sync()
    for i in range(args.warmup):
        step(i)
    if distributed:
        # the first collective of a process group pays for RCCL's lazy set-up (channels, kernels): take it out of the timed region
        for gth in gathers:
            gth.all_gather(FB_RING0)
            gth.all_gather(FB_RING1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, last=(i == args.steps - 1))
    enqueue_s = time.perf_counter() - t0      # host time to enqueue K steps (no synchronisation inside)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: the last step's records (and, N > 1, what the gather delivered of them); every
    # context's last output is the same list (same pair, same features)
    last_i = args.steps - 1
    last_c, last_j = last_i % nctx, last_i // nctx
    out = ctxs[last_c].featbuf_download(out_buffer(last_j), NFEAT)
    for c in range(nctx):
        nsteps_c = (args.steps - c + nctx - 1) // nctx
        if nsteps_c > 0:
            o = ctxs[c].featbuf_download(out_buffer(nsteps_c - 1), NFEAT)
            assert np.array_equal(o["x"], out["x"]) and np.array_equal(o["y"], out["y"]) and np.array_equal(o["val"], out["val"]), \
                "contexts disagree on the tracked records"
    if distributed:                 # what rank 0 received from itself equals what it produced
        got = gathers[last_c].result()[rank].reshape(GATHER_EVERY, NFEAT)[last_j % GATHER_EVERY]
        assert np.array_equal(got["x"], out["x"]) and np.array_equal(got["val"], out["val"]), "gathered records differ"
    tracked = int(np.count_nonzero(out["val"] >= 0))
    live = out["val"] == 0
    shift = (float(np.median(out["x"][live] - fl["x"][live])), float(np.median(out["y"][live] - fl["y"][live])))

    # second pass: per-kernel HIP-event timing + iteration counters for the roofline
    roofline = None
    kernels = []
    if rank == 0:
        ctx.track_stats_reset()
        ctx.timing_enable(True)
        for i in range(args.steps):
            a = 0 if i % 2 == 0 else 2
            ctx.build_pyramids_batch([a, a + 1])
            ctx.track_async(a, a + 1, FB_SEL, FB_OUT0, NFEAT)
        kernels = ctx.timing_read()
        ctx.timing_enable(False)
        st = ctx.track_stats()
        st = {k: ([x / args.steps for x in v] if isinstance(v, list) else v / args.steps) for k, v in st.items()}
        pyr_bytes, track_bytes = algorithmic_bytes(p, WIDTH, HEIGHT, st, NFEAT)
        for k in kernels:
            if k["name"] == "track":
                k["bytes"] = track_bytes * k["launches"]
        dom = max(kernels, key=lambda k: k["total_ms"])
        per_launch_ms = dom["total_ms"] / dom["launches"]
        per_launch_bytes = dom["bytes"] / dom["launches"]
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom["name"])
        # the same kernel against the roof that actually bounds it: VALU issue.  Wavefront-instructions per launch come from a
        # rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/sq_counters.json, tools/pmc_sq.py); 4.5 clocks per instruction and SIMD
        # is what the FP64-rate instruction mix of the convolutions sustains on gfx950 (tools/mb/valu_rate.hip, fp64_mix.hip).
        issue = None
        spath = os.path.join(ROOT, "profiles", "sq_counters.json")
        if os.path.exists(spath):
            sq = json.load(open(spath)).get(dom["name"])
            if sq and sq.get("SQ_INSTS_VALU"):
                simds, cpi, mhz = 256 * 4, 4.5, 2400.0
                ideal_us = sq["SQ_INSTS_VALU"] / simds * cpi / mhz
                issue = {"valu_wavefront_instructions_per_launch": sq["SQ_INSTS_VALU"], "simds": simds, "clocks_per_instruction": cpi,
                         "clock_mhz": mhz, "ideal_us": ideal_us, "frac": ideal_us / (per_launch_ms * 1e3)}
        dev_ms = sum(k["total_ms"] for k in kernels) / args.steps
        roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "issue_bound": issue,
                    "launch_us": per_launch_ms * 1e3, "launches_per_step": dom["launches"] / args.steps,
                    "algorithmic_bytes_per_launch": per_launch_bytes,
                    "step_algorithmic_bytes": 2 * pyr_bytes + track_bytes,
                    "step_device_ms": dev_ms,
                    "step_frac": (2 * pyr_bytes + track_bytes) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "newton_iterations_per_level": st["iterations"][:p.nPyramidLevels],
                    "kernels": {k["name"]: {"us_per_launch": 1e3 * k["total_ms"] / k["launches"],
                                            "launches_per_step": k["launches"] / args.steps,
                                            "GBps": k["bytes"] / max(k["total_ms"], 1e-9) / 1e6} for k in kernels}}

    # secondary figures (never `value`): selection time, and the PCIe-inclusive pair time
    extra = None
    if rank == 0:
        reps = max(5, min(20, args.steps))
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            ctx.select_async(0, 1, True, FB_OUT1, NFEAT)      # SELECTING_ALL on the resident level-0 pyramid
        ctx.sync()
        ms_select = (time.perf_counter() - t) / reps * 1e3
        t = time.perf_counter()
        for i in range(args.steps):                            # the same K steps on ONE stream (one pair in flight)
            a = 0 if i % 2 == 0 else 2
            ctx.build_pyramids_batch([a, a + 1])
            ctx.track_async(a, a + 1, FB_SEL, FB_OUT0 if i % 2 == 0 else FB_OUT1, NFEAT)
        ctx.sync()
        ms_single = (time.perf_counter() - t) / args.steps * 1e3
        t = time.perf_counter()
        for _ in range(reps):                                  # un-pipelined latency of one pair
            ctx.build_pyramids_batch([0, 1])
            ctx.track_async(0, 1, FB_SEL, FB_OUT0, NFEAT)
            ctx.sync()
        ms_latency = (time.perf_counter() - t) / reps * 1e3
        t = time.perf_counter()
        for _ in range(reps):
            ctx.upload(0, f0)
            ctx.upload(1, f1)
            ctx.build_pyramids_batch([0, 1])
            ctx.track_async(0, 1, FB_SEL, FB_OUT0, NFEAT)
            ctx.featbuf_download(FB_OUT0, NFEAT)
        ms_pcie = (time.perf_counter() - t) / reps * 1e3
        # pipelined ingest: frames already sit in pinned host memory (as a decoder would leave them), uploads run on the
        # copy stream and overlap the previous pair's kernels; records go to a device table read back every 16 pairs
        pins = {s0: ctx.pinned_array((HEIGHT, WIDTH)) for s0 in (0, 1, 2, 3)}
        for s0 in (0, 2):
            pins[s0][:] = f0
            pins[s0 + 1][:] = f1
        TAB, NT = 90, 16
        ctx.featbuf_alloc(TAB, NT * NFEAT)
        for k in range(NT):
            ctx.featbuf_view(TAB + 1 + k, TAB, k * NFEAT, NFEAT)
        npipe = 8 * NT

        def pipelined_step(i):
            a = 0 if i % 2 == 0 else 2
            ctx.upload_async(a, pins[a])
            ctx.upload_async(a + 1, pins[a + 1])
            ctx.build_pyramids_batch([a, a + 1])
            ctx.track_async(a, a + 1, FB_SEL, TAB + 1 + i % NT, NFEAT)
            return ctx.featbuf_download(TAB, NT * NFEAT) if i % NT == NT - 1 else None

        for i in range(NT):                 # warm-up: the alternate raw buffers are allocated on first use
            table = pipelined_step(i)
        ctx.sync()
        t = time.perf_counter()
        for i in range(npipe):
            got = pipelined_step(i)
            table = got if got is not None else table
        ctx.sync()
        ms_pipe = (time.perf_counter() - t) / npipe * 1e3
        assert np.array_equal(table[-NFEAT:]["x"], out["x"]), "pipelined ingest changed the result"
        extra = {"pcie_pipelined_ms_per_pair": ms_pipe, "pcie_pipelined_features_per_s": NFEAT / (ms_pipe * 1e-3),
                 "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3, "latency_ms_per_pair_synchronised": ms_latency,
                 "single_stream_ms_per_pair": ms_single, "single_stream_features_per_s": NFEAT / (ms_single * 1e-3),
                 "ms_per_select_5000": ms_select,
                 "pcie_inclusive_ms_per_pair": ms_pcie, "pcie_inclusive_features_per_s": NFEAT / (ms_pcie * 1e-3),
                 "note": "pcie_inclusive = H2D of two u8 frames from pageable host memory + pyramids + track + D2H of the "
                         "records, synchronised per pair; pcie_pipelined = the same bytes with klt_upload_u8_async from "
                         "pinned memory on a copy stream and the records read back every 16 pairs"}

    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        cpu = cpu_baseline(p, f0, f1, fl)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "features tracked/sec", "value": world * NFEAT * args.steps / elapsed, "unit": "features/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_frame_pair": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (convolutions accumulate in f64)", "data": "synthetic",
            "config": {"workload": "cfg-2: one 1920x1080 synthetic pair per GPU, 5000 features, 7x7 window, "
                                   "3 pyramid levels (subsampling 4), translation only; inputs resident in HBM",
                       "pipelining": ("none (one HIP stream)" if nctx == 1 else
                                      "%d independent pairs in flight: steps go round-robin to %d contexts, one HIP stream each, no "
                                      "ordering between them (pairs are independent); every step does the full work of one pair" % (nctx, nctx))
                                     + ("; tracker of a step on its own HIP stream (KLT_OPT_TRACK_STREAM)" if args.pipeline else ""),
                       "pairs_in_flight": nctx,
                       "features_per_pair": NFEAT, "pairs_per_step": world, "tracked": tracked,
                       "recovered_shift_px": shift, "imposed_shift_px": list(synth.DEFAULT_SHIFT),
                       "parallelism": "1 pair per GPU" + (", RCCL all-gather of the [%d steps x 5000] record table every %d steps" % (GATHER_EVERY, GATHER_EVERY) if distributed else "")},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
    else:
        line = None
    for cx in ctxs:
        cx.close()
    if distributed:
        dist.destroy_process_group()        # RCCL may write its own chatter to stdout while shutting down
    if line is not None:
        os.write(json_fd, (json.dumps(line) + "\n").encode())      # the ONE JSON line on the real stdout


if __name__ == "__main__":
    main()
