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
every rank with one RCCL all-gather issued by libkltgpu.so on its side stream (event-ordered behind the
tracker launch, overlapped with the next steps' kernels).  No torch anywhere.

Launching.  `--gpus N` with N > 1 and no RANK in the environment: this process -- before it touches
the GPU in any way -- starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous
file in their environment), forwards rank 0's JSON line and exits non-zero if any rank failed.  Under an
external launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) the same
variables are already set and every process is a rank.  The RCCL unique id travels through the
rendezvous file (pyfeaturetrack_amd/parallel.py).

Consecutive groups of `--batch` steps (default 2) go round-robin to `--inflight` contexts (default 2; one HIP stream each, nothing
ordering them): frame pairs are independent, so the steps of a group share every launch of their context (one batched pyramid build,
one tracker launch -- the reference's workload for a stereo rig or two cameras) and the GPU overlaps the kernels of different groups.
Every step does the full work of one pair; `ms_per_frame_pair` (= `extra.single_stream_ms_per_pair`) is one pair at a time on one
stream, `--inflight 3 --batch 1` the round-1 arrangement.  The K-step timed region (barrier +
synchronise on both sides, MAX over ranks) is repeated `--repeats` times; `ms_per_step` is the median region, the spread is
in `extra.region_ms_per_step`.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  parity_checked -- the records of the timed steps equal the CPU oracle's on the same pair (the run fails otherwise);
  roofline     -- the dominant kernel of the step (largest share of device time), timed with HIP
                  events on the context's stream in a second pass over the same K steps (events
                  around every launch would distort the un-instrumented `value`);
  cpu_baseline -- the CPU oracle (oracle/klt_oracle.c, a bit-exact port of the reference's
                  Python/Cython/SciPy path) on the same workload, 1 thread, rank 0, N = 1 only.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# importing the package does not touch the GPU (the library is bound on first use)
from pyfeaturetrack_amd import parallel, synth                          # noqa: E402
from pyfeaturetrack_amd.backend import Context                          # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                  # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc                    # noqa: E402

WIDTH, HEIGHT, NFEAT = 1920, 1080, 5000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TOL_PX = 1e-3             # north_star: sub-pixel x/y within 1e-3 (observed: 0)
FB_SEL, FB_OUT0, FB_OUT1 = 0, 1, 2
FB_RING0, FB_RING1, FB_GATH0, FB_GATH1, FB_VIEW0 = 3, 4, 5, 6, 100
GATHER_EVERY = 128
DTYPE = "f32 (convolutions accumulate in f64)"


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


def oracle_track(p, f0, f1, fl):
    """The CPU oracle's records for one pair (the checker; never the thing measured).  None if the oracle is not built."""
    try:
        from oracle import klt_oracle as ko
    except (ImportError, OSError) as e:
        print("oracle unavailable: %s" % e, file=sys.stderr)
        return None
    a0, a1 = f0.astype(np.float32), f1.astype(np.float32)
    ofl = fl.copy()
    ko.set_threads(1)
    ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), ofl)
    return ofl


def parity_against(out, ofl):
    """{parity_checked, max_abs_dx, ...} of timed records `out` against the oracle's `ofl` (None: unchecked)."""
    if ofl is None:
        return {"parity_checked": False, "parity_note": "oracle library not built on this box"}
    same_val = bool(np.array_equal(out["val"], ofl["val"]))
    dx = float(max(np.abs(out["x"].astype(np.float64) - ofl["x"]).max(), np.abs(out["y"].astype(np.float64) - ofl["y"]).max()))
    return {"parity_checked": bool(same_val and dx <= TOL_PX), "max_abs_dx": dx, "status_codes_equal": same_val,
            "parity_tolerance_px": TOL_PX, "parity_against": "oracle/klt_oracle.c (pinned to reference-generated goldens)"}


def cpu_baseline(p, f0, f1, fl, nfeat, label):
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
    reps = int(max(2, min(200, 10.0 / max(t1, 1e-3))))      # about 10 s of single-thread work
    t = time.perf_counter()
    for _ in range(reps):
        one_pair()
    dt = (time.perf_counter() - t) / reps
    # the same port on the host cores this process may actually use (OpenMP over image lines / features;
    # bit-identical results).  Time-bounded: a container with a CPU quota can make many threads slower than one.
    ncores = ko.set_threads(usable_cores())
    t = time.perf_counter()
    reps_all = 0
    while reps_all < 40 and (reps_all < 2 or time.perf_counter() - t < 4.0) and time.perf_counter() - t < 12.0:
        one_pair()
        reps_all += 1
    dt_all = (time.perf_counter() - t) / reps_all
    ko.set_threads(1)
    return {"value": nfeat / dt, "unit": "features/s", "cores": 1, "kind": "port",
            "ms_per_pair": dt * 1e3,
            "sample": "%d x (pyramids of both frames + track %d features) of %s, oracle/klt_oracle.c, 1 thread" % (reps, nfeat, label),
            "all_cores": {"value": nfeat / dt_all, "cores": ncores, "ms_per_pair": dt_all * 1e3,
                          "sample": "%d x the same pair, OpenMP over image lines and features" % reps_all}}


def file_sha16(rel):
    try:
        return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def committed_counters(name, kernel_family, pairs_per_launch=None):
    """Per-launch PMC figures of `kernel_family` from profiles/<name> -- NOT measurements of this run: they come from the
    builder's rocprofv3 --pmc passes (tools/pmc_traffic.py, tools/pmc_sq.py) and carry their provenance; they are dropped when
    the kernel source they were collected for is no longer the one in the tree."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    data = json.load(open(path))
    meta = data.get("_meta", {})
    for rel, sha in (meta.get("kernel_source_sha16") or {}).items():
        if file_sha16(rel) != sha:
            return None, "profiles/%s is stale: %s changed since it was collected" % (name, rel)
    if not meta:
        return None, "profiles/%s carries no provenance record" % name
    if pairs_per_launch is not None and meta.get("cfg2_pairs_per_launch", 1) != pairs_per_launch:
        return None, "profiles/%s was collected for launches of %d pair(s), this run's hold %d" % (name, meta.get("cfg2_pairs_per_launch", 1), pairs_per_launch)
    return data.get(kernel_family), "profiles/%s, %s" % (name, meta.get("source", "builder gpurun"))


# ================================================================================== timing helpers
class Ranks:
    """Rank bookkeeping + the barrier / max-over-ranks of the timing contract, through libkltgpu's RCCL entry points."""

    def __init__(self, args):
        self.rank, self.local_rank, self.world = parallel.world_from_env()
        if os.environ.get("KLT_RANKS_SHARE_DEVICE") is not None:      # test hook: several ranks on one GPU (a one-GPU box)
            self.local_rank = int(os.environ["KLT_RANKS_SHARE_DEVICE"])
        if self.world != args.gpus and self.world > 1:
            print("warning: WORLD_SIZE=%d but --gpus %d" % (self.world, args.gpus), file=sys.stderr)
        self.distributed = self.world > 1 or os.environ.get("KLT_FORCE_DIST") == "1"   # the env var exercises the RCCL path on one GPU
        self.ctxs = []

    def attach(self, ctxs):
        """One communicator per context, same order on every rank."""
        self.ctxs = list(ctxs)
        if self.distributed:
            parallel.init_communicators(self.ctxs, self.rank, self.world)
            self.max_over_ranks(0.0)                      # first collective: RCCL's lazy set-up, and every rank has joined
            parallel.cleanup_rendezvous(self.rank)

    def sync_local(self):
        for cx in self.ctxs:
            cx.sync()                                      # stream + copy stream + this context's collectives

    def max_over_ranks(self, v):
        if not self.distributed:
            return v
        return self.ctxs[0].comm_allreduce_max([float(v)])[0]

    def fence(self):
        """everything enqueued so far has finished on every rank"""
        self.sync_local()
        self.max_over_ranks(0.0)

    def timed(self, fn):
        """fence; run fn(); synchronise; elapsed seconds = MAX over ranks"""
        self.fence()
        t0 = time.perf_counter()
        fn()
        enq = time.perf_counter() - t0
        self.sync_local()
        el = time.perf_counter() - t0
        return self.max_over_ranks(el), enq


def timed_regions(ranks, run_region, steps, repeats, budget_s=25.0):
    """`repeats` K-step regions (each bracketed as the contract says).  Returns (median seconds per region, all regions, host
    enqueue seconds of the median region).  The repeat count shrinks (never below 5) if the regions are long."""
    el, enq = ranks.timed(run_region)
    regions = [(el, enq)]
    n = repeats
    if el * repeats > budget_s:
        n = max(5, int(budget_s / max(el, 1e-9)))
    n = int(ranks.max_over_ranks(n)) if ranks.distributed else n      # every rank runs the same number of regions
    while len(regions) < n:
        regions.append(ranks.timed(run_region))
    regions.sort()
    med = regions[len(regions) // 2]
    return med[0], [r[0] for r in regions], med[1]


def emit(json_fd, line):
    os.write(json_fd, (json.dumps(line) + "\n").encode())      # the ONE JSON line on the real stdout


def base_line(value, n_gpus, steps, warmup, ms_step, ms_pair, workload, scaling="weak", extra_cfg=None):
    cfg = {"workload": workload}
    cfg.update(extra_cfg or {})
    return {"metric": "features tracked/sec", "value": value, "unit": "features/s", "n_gpus": n_gpus, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_step, "ms_per_frame_pair": ms_pair, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": cfg, "roofline": None, "cpu_baseline": None}


# ========================================================================================== cfg-4
def run_cfg4(args, json_fd):
    """BASELINE cfg-4: 256 independent 1280x720 pairs (seeds 0..255), 2000 features each, 7x7, 3 levels / ss 4, sharded
    contiguously over the ranks (32 per GPU at N = 8), frames resident in HBM.  Per step every rank builds the pyramids
    of its whole shard (frames share launches through blockIdx.z), tracks it with ONE launch into a device-side
    [pairs x features] table and the table is gathered to rank 0 with one RCCL gather.  Total work is fixed: strong scaling."""
    ranks = Ranks(args)
    total, w, h, nf = args.pairs, 1280, 720, 2000
    mine = parallel.shard_range(total, ranks.world, ranks.rank)
    pairs = len(mine)
    if ranks.distributed and total % ranks.world:
        raise SystemExit("--pairs must be a multiple of the number of ranks")
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
    ctx.build_pyramids_batch(slots, sync=True)
    T_IN, T_OUT, T_ALL, V_IN, V_OUT = 0, 1, 2, 1000, 1000 + pairs
    ctx.featbuf_alloc(T_IN, pairs * nf)
    ctx.featbuf_alloc(T_OUT, pairs * nf)
    for k in range(pairs):
        ctx.featbuf_view(V_IN + k, T_IN, k * nf, nf)
        ctx.featbuf_view(V_OUT + k, T_OUT, k * nf, nf)
        ctx.select_async(2 * k, 1, True, V_IN + k, nf)
    ctx.sync()
    table = [(2 * k, 2 * k + 1, V_IN + k, V_OUT + k) for k in range(pairs)]
    ranks.attach([ctx])
    gather = parallel.ShardGather(ctx, T_OUT, T_ALL, pairs, nf, root=0) if ranks.distributed else None

    def step():
        ctx.build_pyramids_batch(slots)
        if gather:
            ctx.comm_fence_featbuf(T_OUT)          # the gather of the previous step has read the table
        ctx.track_batch_async(table, nf)
        if gather:
            gather.gather_async()

    def region():
        for _ in range(args.steps):
            step()

    for _ in range(max(1, args.warmup)):
        step()
    el, regions, enq = timed_regions(ranks, region, args.steps, args.repeats)
    # what was timed, against the oracle: the first pair of rank 0's shard
    out = ctx.featbuf_download(T_OUT, pairs * nf).reshape(pairs, nf)
    fl0 = ctx.featbuf_download(V_IN, nf)
    par = parity_against(out[0], oracle_track(p, frames[0][0], frames[0][1], fl0)) if ranks.rank == 0 else {}
    gathered_ok = None
    if gather:
        full = gather.result()
        if ranks.rank == 0:
            gathered_ok = bool(full.shape == (total, nf) and np.array_equal(full[:pairs], out))
    roof = None
    if ranks.rank == 0:
        ctx.track_stats_reset()
        ctx.timing_enable(True)
        for _ in range(min(args.steps, 20)):
            ctx.build_pyramids_batch(slots)
            ctx.track_batch_async(table, nf)
        kernels = ctx.timing_read()
        ctx.timing_enable(False)
        nst = min(args.steps, 20)
        st = ctx.track_stats()
        st = {k: ([x / (nst * pairs) for x in v] if isinstance(v, list) else v / (nst * pairs)) for k, v in st.items()}
        pyr_bytes, track_bytes = algorithmic_bytes(p, w, h, st, nf)
        step_bytes = total * (2 * pyr_bytes + track_bytes)
        roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS * ranks.world, "step_algorithmic_bytes": step_bytes,
                "achieved": step_bytes / el * args.steps / 1e9, "frac": step_bytes / el * args.steps / 1e9 / (HBM_PEAK_GBS * ranks.world),
                "traffic": None,
                "kernels_rank0": {k["name"]: {"us_per_launch": 1e3 * k["total_ms"] / k["launches"],
                                              "launches_per_step": k["launches"] / nst} for k in kernels}}
    ctx.close()
    if ranks.rank == 0:
        ms_step = el / args.steps * 1e3
        tracked = int(np.count_nonzero(out["val"] >= 0))
        line = base_line(total * nf * args.steps / el, ranks.world, args.steps, args.warmup, ms_step, ms_step / total,
                         "cfg-4: %d independent 1280x720 pairs per step (%d per GPU), 2000 features each, 7x7, 3 levels "
                         "(subsampling 4); per rank: batched pyramid build + one tracker launch + one RCCL gather of the "
                         "[pairs x 2000] record table to rank 0" % (total, pairs), scaling="strong",
                         extra_cfg={"pairs_per_step": total, "pairs_per_rank": pairs, "tracked_rank0": tracked,
                                    "rccl_ranks": ranks.world if ranks.distributed else 0, "gathered_table_ok": gathered_ok,
                                    "parallelism": "pairs sharded contiguously, %d per GPU; no data-path collective, one gather" % pairs})
        line.update(par)
        line["roofline"] = roof
        line["extra"] = {"region_ms_per_step": {"median": ms_step, "min": min(regions) / args.steps * 1e3,
                                                "max": max(regions) / args.steps * 1e3, "regions": len(regions)},
                         "host_enqueue_ms_per_step": enq / args.steps * 1e3}
        emit(json_fd, line)
        if par and not par.get("parity_checked") and "max_abs_dx" in par:
            raise SystemExit("timed records differ from the oracle: %r" % par)


# ==================================================================================== cfg-1 / 3 / 5
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
    line = base_line(100 * args.steps / el, 1, args.steps, args.warmup, el / args.steps * 1e3, el / args.steps * 1e3,
                     "cfg-1: img0.pgm -> img1.pgm (320x240), 100 features, 7x7, 2 levels (ss 4), max_residue 10",
                     extra_cfg={"tracked": int((out["val"] >= 0).sum()), "ms_select_100": ms_select})
    line.update(parity_against(out, oracle_track(params_from_tc(tc), i0, i1, fl)))
    emit(json_fd, line)


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
    emit(json_fd, base_line(live1 * args.steps / el, 1, args.steps, args.warmup, el / args.steps * 1e3, el / args.steps * 1e3,
                            "cfg-3: 1920x1080, %d features placed (%d live), 15x15 window, 4 levels (ss 2), affine consistency "
                            "check mode 2 (parity unpinned); per step: pyramids of both frames + translation tracker + affine check"
                            % (placed, live1),
                            extra_cfg={"tracked_after_affine": int((out["val"] >= 0).sum()), "kernel_us": kern}))


def run_cfg5(args, json_fd):
    """BASELINE cfg-5 (single GPU): 3840x2160 sequence, 20000 features, sequential mode, lost features replaced after every
    frame.  Per step: upload is excluded (frames resident), pyramid of the new frame, track, REPLACING_SOME selection."""
    ranks = Ranks(args)
    if ranks.distributed:
        return run_cfg5_blocks(args, json_fd, ranks)
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

    # the pyramids of frame k+1 are built on the context's build stream while frame k is tracked and its lost features are replaced
    # (KLT_OPT_BUILD_STREAM; same results -- every frame has its own slot here)
    prefetch = os.environ.get("KLT_BENCH_NO_PREFETCH") != "1"
    prepare = prefetch and os.environ.get("KLT_BENCH_NO_PREPARE") != "1"
    if prefetch:
        ctx.set_option(15, 1)

    lost = []

    def run_sequence(timed):
        t_sel = 0.0
        if prefetch:
            ctx.build_pyramids(10 + 1, sync=False)
            if prepare:
                ctx.select_prepare(10 + 1)
        for k in range(1, nframes):
            if not prefetch:
                ctx.build_pyramids(10 + k, sync=False)
            ctx.track_async(10 + k - 1, 10 + k, (k - 1) % 2, k % 2, n)      # the chain first: the build stream waits for nothing on this one
            if prefetch and not timed:
                ctx.select_begin(10 + k, 2, True, k % 2, n)   # ... KLTReplaceLostFeatures on the resident level-0 images, up to the host's look
            if prefetch and k + 1 < nframes:
                ctx.build_pyramids(10 + k + 1, sync=False)
                if prepare:
                    ctx.select_prepare(10 + k + 1)        # SAT + eigenvalues of the next frame, behind its build on the build stream
            if prefetch and not timed:
                ctx.select_finish()
                continue
            if timed:
                lost.append(int((ctx.featbuf_download(k % 2, n)["val"] < 0).sum()))      # (synchronises)
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
    emit(json_fd, base_line(n * frames_done / el, 1, frames_done, 0, el / frames_done * 1e3, el / frames_done * 1e3,
                            "cfg-5 (one GPU): 3840x2160 sequence, 20000 features, 7x7, 3 levels (ss 4), sequential mode, lost "
                            "features replaced after every frame; per frame: pyramid of the new frame + track + replacement"
                            + ("; the next frame's pyramids are built on a second stream meanwhile" if prefetch else "")
                            + (", and so are its summed-area tables and eigenvalues (klt_select_prepare_async)" if prepare else ""),
                            extra_cfg={"live_at_end": int((out["val"] >= 0).sum()), "ms_replace_per_frame": t_sel / (nframes - 1) * 1e3,
                                       "lost_per_frame": lost, "build_stream": bool(prefetch), "scores_prepared": bool(prepare)}))


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
    el, regions, enq = timed_regions(ranks, block, B * world, max(5, min(args.repeats, 15)))
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
                                extra_cfg={"rccl_ranks": world, "live_after_each_block": [int((t["val"] >= 0).sum()) for t in table],
                                           "ms_per_frame_of_the_chain": el / (reps * B * world) * 1e3, "baton_copy_ok": baton_ok,
                                           "region_ms": {"median": el * 1e3, "min": min(regions) * 1e3, "max": max(regions) * 1e3}}))
    ctx.close()


# ================================================================================= launcher dry run
def dry_run(args, json_fd):
    """KLT_BENCH_DRYRUN=1: launcher + rendezvous + shard arithmetic without a GPU (the collective is stubbed by files).
    Exercised by the CPU tests with 2 processes."""
    rank, local_rank, world = parallel.world_from_env()
    path = parallel.rendezvous_file()
    ids = parallel.exchange_ids(rank, world, 3, lambda: os.urandom(parallel.KLT_COMM_ID_BYTES), path=path, timeout=60)
    digest = hashlib.sha256(b"".join(ids)).hexdigest()
    mine = list(parallel.shard_range(args.pairs, world, rank))
    if os.environ.get("KLT_DRYRUN_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    with open("%s.rank%d" % (path, rank), "w") as f:
        json.dump({"digest": digest, "pairs": mine, "local_rank": local_rank}, f)
    if rank != 0:
        return
    seen = []
    t0 = time.monotonic()
    for r in range(world):
        while True:
            try:
                seen.append(json.load(open("%s.rank%d" % (path, r))))
                break
            except (OSError, ValueError):
                if time.monotonic() - t0 > 60:
                    raise SystemExit("rank %d never reported" % r)
                time.sleep(0.01)
    emit(json_fd, {"dryrun": True, "n_gpus": world, "ids_agree": all(s["digest"] == digest for s in seen),
                   "pairs_covered": sorted(i for s in seen for i in s["pairs"]) == list(range(args.pairs)),
                   "local_ranks": [s["local_rank"] for s in seen], "spawned": os.environ.get("KLT_SPAWNED") == "1"})


# ============================================================================================ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=25,
                    help="how many times the K-step timed region is run (median reported; fewer, never below 5, when a region is long)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed hot-path work before the W warm-up steps: the GPU needs ~10 ms of load to reach its steady clocks / "
                         "cache state (a 200-step run right after start-up measures 50 us per pair, the same loop after 50 ms 43 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures (selection, one pair at a time, PCIe-inclusive): a profiler then sees only the launches "
                         "of the timed regions and of the roofline pass, all of the headline's size")
    ap.add_argument("--config", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the headline line); cfg4 = the 256-pair batch sharded over --gpus ranks; the others are "
                         "the remaining BASELINE configs on one GPU, informative")
    ap.add_argument("--pairs", type=int, default=256, help="total pairs per step for --config cfg4 (sharded over the ranks)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="contexts per GPU (one HIP stream each, no events between them): consecutive groups of --batch steps go "
                         "round-robin to them, so kernels of different groups overlap; 1 = a single stream")
    ap.add_argument("--slot-sets", type=int, default=1, choices=[1, 2],
                    help="1 (default): a context rebuilds the same slots every group, as a caller with a fixed ring of frame slots does -- the "
                         "pyramids the tracker reads are the ones just written and come out of the 256 MB Infinity Cache; 2: alternate groups "
                         "use two sets of slots (the working set of 2 contexts x 2 pairs is then 430 MB and every tracker read goes to HBM; "
                         "the lines up to round-2 set m were measured this way)")
    ap.add_argument("--batch", type=int, default=2, choices=[1, 2, 4, 8],
                    help="steps (independent pairs) that share every launch of a context: one batched pyramid build for their frames "
                         "and one tracker launch for their feature lists (2 contexts x 2 pairs: 0.0333 ms per pair where 3 x 1 reads "
                         "0.0377, 3 x 2 0.0358, 2 x 4 0.0359: tools/stream_batch_probe.py)")
    args = ap.parse_args()

    # N > 1 without a launcher: start the ranks ourselves.  Nothing above or below this point has touched the GPU yet
    # (no HIP call, no library load): the children are fresh processes, this one only waits for them.
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(parallel.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    # stdout must carry exactly one JSON line.  RCCL / the HIP runtime print their own chatter to fd 1 (also at
    # process exit), so fd 1 is pointed at stderr for the whole run and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("KLT_BENCH_DRYRUN") == "1":
        return dry_run(args, json_fd)
    if args.config != "cfg2":
        return {"cfg1": run_cfg1, "cfg3": run_cfg3, "cfg4": run_cfg4, "cfg5": run_cfg5}[args.config](args, json_fd)

    ranks = Ranks(args)
    rank, world, distributed = ranks.rank, ranks.world, ranks.distributed
    tc = cfg2_context()
    p = params_from_tc(tc)
    f0, f1 = synth.synth_pair(WIDTH, HEIGHT, seed=rank + 1)
    # `--inflight` contexts per GPU, each with its own HIP stream, slots and feature buffers.  Step i runs on context
    # i % inflight: pairs are independent (the path shards by frame pair), so nothing orders the streams against each other
    # and the GPU overlaps the kernels of different pairs -- the drain / ramp between dependent kernels of one pair and the
    # latency-bound tracker are filled with the next pair's convolutions.
    nctx = max(1, args.inflight)
    B = max(1, args.batch)
    NSETS = args.slot_sets

    def slots_of(j, nb=None):
        """frame slots of a context's j-th group: pair b of set t lives in slots 2 (t B + b), + 1"""
        t = j % NSETS
        return [2 * (t * B + b) + f for b in range(B if nb is None else nb) for f in (0, 1)]

    def pairs_of(j, outs, nb=None):
        sl = slots_of(j, nb)
        return [(sl[2 * b], sl[2 * b + 1], FB_SEL, outs[b]) for b in range(len(sl) // 2)]

    def plain_out(t, b):
        """output buffer of pair b of set t (one GPU): the two of pair 0 are FB_OUT0 / FB_OUT1"""
        return (FB_OUT0, FB_OUT1)[t] if b == 0 else 10 + 2 * b + t

    ctxs = []
    fl = None
    for c in range(nctx):
        cx = Context(ranks.local_rank)
        cx.set_params(p)
        for s0 in range(0, max(4, 2 * NSETS * B), 2):
            cx.upload(s0, f0)
            cx.upload(s0 + 1, f1)
        cx.build_pyramids(0)
        fl_c, placed = cx.select(0, NFEAT, use_pyramid=True)
        assert placed == NFEAT, "only %d of %d features could be placed" % (placed, NFEAT)
        assert fl is None or np.array_equal(fl, fl_c), "contexts selected different features"
        fl = fl_c
        cx.featbuf_upload(FB_SEL, fl)
        for t in (0, 1):
            for b in range(B):
                cx.featbuf_upload(plain_out(t, b), fl)
        # N > 1: the records of GATHER_EVERY consecutive steps of a context land in one device-side [steps x features] table
        # (two tables, used alternately) and each full table is all-gathered with ONE RCCL collective on the library's side
        # stream -- cfg-4's "gather once per shard", and the host cost of a collective is not paid per step.
        if distributed:
            for t, ring in enumerate((FB_RING0, FB_RING1)):
                cx.featbuf_alloc(ring, GATHER_EVERY * NFEAT)
                for k in range(GATHER_EVERY):
                    cx.featbuf_view(FB_VIEW0 + t * GATHER_EVERY + k, ring, k * NFEAT, NFEAT)
        ctxs.append(cx)
    ctx = ctxs[0]
    ranks.attach(ctxs)

    def out_buffer(lj):
        """feature buffer that the lj-th pair of a context writes (lj = group * B + pair)"""
        if not distributed:
            return plain_out((lj // B) % 2, lj % B)
        return FB_VIEW0 + ((lj // GATHER_EVERY) % 2) * GATHER_EVERY + lj % GATHER_EVERY

    def group_build(g, nb=B):
        """nb consecutive steps (pairs) as one group: every launch of the context is shared by them.  First half: the pyramids"""
        c, j = g % nctx, g // nctx                # context, and the group's index among that context's groups
        ctxs[c].build_pyramids_batch(slots_of(j, nb))            # all frames of the group share every launch

    def group_track(g, nb=B, last=False):
        """second half: the tracker launch (and, N > 1, the collective behind it)"""
        c, j = g % nctx, g // nctx
        cx = ctxs[c]
        slots = slots_of(j, nb)
        lj0 = j * B
        if distributed and lj0 % GATHER_EVERY == 0:
            cx.comm_fence_featbuf(FB_RING0 if (lj0 // GATHER_EVERY) % 2 == 0 else FB_RING1)   # the collective that read this table two rounds ago has finished
        if nb == 1:
            cx.track_async(slots[0], slots[1], FB_SEL, out_buffer(lj0), NFEAT)
        else:
            cx.track_batch_async(pairs_of(j, [out_buffer(lj0 + b) for b in range(nb)], nb), NFEAT)
        if distributed and ((lj0 + B) % GATHER_EVERY == 0 or last):
            ring, gath = (FB_RING0, FB_GATH0) if (lj0 // GATHER_EVERY) % 2 == 0 else (FB_RING1, FB_GATH1)
            cx.allgather_featbuf_async(ring, gath, GATHER_EVERY * NFEAT)     # RCCL on the side stream, behind this tracker launch

    def run_steps(n):
        """n steps = ceil(n / B) groups, the last one partial when B does not divide n.  The groups go out in rounds of one group per
        context, the builds of a round before its tracker launches: every stream has work a few microseconds after the region starts
        (enqueueing a whole group takes the host ~25 us); the order inside each stream, and the work, are the same either way"""
        ngroups = (n + B - 1) // B
        for g0 in range(0, ngroups, nctx):
            wave = range(g0, min(g0 + nctx, ngroups))
            for g in wave:
                group_build(g, nb=min(B, n - g * B))
            for g in wave:
                group_track(g, nb=min(B, n - g * B), last=(g >= ngroups - nctx))     # every context closes its open table with a gather

    # bring the GPU to its steady state first (the same work as the steps, into the plain output buffers)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for g in range(32):
            cx, j = ctxs[g % nctx], g // nctx
            cx.build_pyramids_batch(slots_of(j))
            if B == 1:
                cx.track_async(slots_of(j)[0], slots_of(j)[1], FB_SEL, plain_out(j % 2, 0), NFEAT)
            else:
                cx.track_batch_async(pairs_of(j, [plain_out(j % 2, b) for b in range(B)]), NFEAT)
        for cx in ctxs:
            cx.sync()
    if args.warmup:
        run_steps(args.warmup)

    def region():
        run_steps(args.steps)

    elapsed, regions, enqueue_s = timed_regions(ranks, region, args.steps, args.repeats)

    # correctness of what was timed: the last step's records (and, N > 1, what the gather delivered of them); every
    # context's last output is the same list (same pair, same features)
    def where(i):
        """(context, index among the context's pairs) of step i"""
        g = i // B
        return g % nctx, (g // nctx) * B + i % B

    last_c, last_lj = where(args.steps - 1)
    out = ctxs[last_c].featbuf_download(out_buffer(last_lj), NFEAT)
    for i in range(max(0, args.steps - 2 * nctx * B), args.steps):          # the last outputs of every context, every pair of a group
        c_i, lj_i = where(i)
        o = ctxs[c_i].featbuf_download(out_buffer(lj_i), NFEAT)
        assert np.array_equal(o["x"], out["x"]) and np.array_equal(o["y"], out["y"]) and np.array_equal(o["val"], out["val"]), \
            "contexts / pairs of a group disagree on the tracked records"
    if distributed:                 # what this rank received from itself equals what it produced
        gath = FB_GATH0 if (last_lj // GATHER_EVERY) % 2 == 0 else FB_GATH1
        got = ctxs[last_c].featbuf_download(gath, world * GATHER_EVERY * NFEAT).reshape(world, GATHER_EVERY, NFEAT)
        mine = got[rank][last_lj % GATHER_EVERY]
        assert np.array_equal(mine["x"], out["x"]) and np.array_equal(mine["val"], out["val"]), "gathered records differ"
    tracked = int(np.count_nonzero(out["val"] >= 0))
    live = out["val"] == 0
    shift = (float(np.median(out["x"][live] - fl["x"][live])), float(np.median(out["y"][live] - fl["y"][live])))
    parity = parity_against(out, oracle_track(p, f0, f1, fl)) if rank == 0 else {}

    # second pass: per-kernel HIP-event timing + iteration counters for the roofline
    roofline = None
    kernels = []
    if rank == 0:
        ngroups_roof = max(1, args.steps // B)
        npairs_roof = ngroups_roof * B

        def roof_groups(n):
            for g in range(n):
                ctx.build_pyramids_batch(slots_of(g))
                if B == 1:
                    ctx.track_async(slots_of(g)[0], slots_of(g)[1], FB_SEL, plain_out(g % 2, 0), NFEAT)
                else:
                    ctx.track_batch_async(pairs_of(g, [plain_out(g % 2, b) for b in range(B)]), NFEAT)

        def roof_pass(mode):
            """the groups of the timed region once more, on one context, with every launch timed -- at the clocks the timed regions ran
            at: the parity check and the downloads above left the GPU idle, so the same untimed groups run first, as before the regions"""
            t_warm = time.perf_counter()
            while (time.perf_counter() - t_warm) * 1e3 < min(args.prewarm_ms, 30.0):
                roof_groups(16)
                ctx.sync()
            ctx.timing_enable(mode)
            for g in range(ngroups_roof):
                ctx.build_pyramids_batch(slots_of(g))
                if B == 1:
                    ctx.track_async(slots_of(g)[0], slots_of(g)[1], FB_SEL, plain_out(g % 2, 0), NFEAT)
                else:
                    ctx.track_batch_async(pairs_of(g, [plain_out(g % 2, b) for b in range(B)]), NFEAT)
            res = ctx.timing_read()
            ctx.timing_enable(False)
            return res

        ctx.track_stats_reset()
        kernels = roof_pass(1)
        st = ctx.track_stats()
        st = {k: ([x / npairs_roof for x in v] if isinstance(v, list) else v / npairs_roof) for k, v in st.items()}
        pyr_bytes, track_bytes = algorithmic_bytes(p, WIDTH, HEIGHT, st, NFEAT)
        for k in kernels:
            if k["name"] == "track":
                k["bytes"] = track_bytes * B * k["launches"]
        dom = max(kernels, key=lambda k: k["total_ms"])
        pair_launch_ms = dom["total_ms"] / dom["launches"]        # an event pair AROUND the launch: the kernel + the boundary to the launch before it
        per_launch_ms = pair_launch_ms
        per_launch_bytes = dom["bytes"] / dom["launches"]
        # the dominant kernel once more, timed by the start / stop events of its own dispatch (klt_timing_enable(ctx, 2): the runtime fills
        # them from the dispatch packet's begin / end timestamps) -- the duration rocprofv3 reports for it, which the event pair above
        # overstates by the ~2.6 us between two dependent launches
        stamp_launch_ms = None
        if dom["name"] == "smooth_grad_l0":
            for k in roof_pass(2):
                if k["name"] == dom["name"] and k["launches"]:
                    stamp_launch_ms = k["total_ms"] / k["launches"]
            if stamp_launch_ms:
                per_launch_ms = stamp_launch_ms
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        # PMC-derived figures are NOT measured by this run: committed results of the builder's rocprofv3 --pmc passes, with their
        # provenance, dropped when the kernel source changed since (committed_counters)
        traffic, traffic_source = committed_counters("traffic.json", dom["name"], B)
        # the same kernel against the roof that actually bounds it: VALU issue.  Wavefront-instructions per launch come from a
        # rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/sq_counters.json, tools/pmc_sq.py); 4.5 clocks per instruction and SIMD
        # is what the FP64-rate instruction mix of the convolutions sustains on gfx950 (tools/mb/valu_rate.hip, fp64_mix.hip).
        issue = None
        sq, sq_source = committed_counters("sq_counters.json", dom["name"], B)
        if sq and sq.get("SQ_INSTS_VALU"):
            simds, cpi, mhz = 256 * 4, 4.5, 2400.0
            ideal_us = sq["SQ_INSTS_VALU"] / simds * cpi / mhz
            issue = {"valu_wavefront_instructions_per_launch": sq["SQ_INSTS_VALU"], "simds": simds, "clocks_per_instruction": cpi,
                     "clock_mhz": mhz, "ideal_us": ideal_us, "frac": ideal_us / (per_launch_ms * 1e3), "source": sq_source}
        elif sq_source:
            issue = {"source": sq_source}
        dev_ms = sum(k["total_ms"] for k in kernels) / npairs_roof
        roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "issue_bound": issue,
                    "launch_us": per_launch_ms * 1e3, "launch_us_source": "dispatch start/stop events (hipExtLaunchKernelGGL)" if stamp_launch_ms
                    else "event pair around the launch", "launch_us_event_pair": pair_launch_ms * 1e3,
                    "launches_per_step": dom["launches"] / npairs_roof, "pairs_per_launch": B,
                    "algorithmic_bytes_per_launch": per_launch_bytes,
                    "step_algorithmic_bytes": 2 * pyr_bytes + track_bytes,
                    "step_device_ms": dev_ms,
                    "step_frac": (2 * pyr_bytes + track_bytes) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "newton_iterations_per_level": st["iterations"][:p.nPyramidLevels],
                    "kernels": {k["name"]: {"us_per_launch": 1e3 * k["total_ms"] / k["launches"],
                                            "launches_per_step": k["launches"] / npairs_roof,
                                            "GBps": k["bytes"] / max(k["total_ms"], 1e-9) / 1e6} for k in kernels}}

    # secondary figures (never `value`): selection time, the one-stream figure, and the PCIe-inclusive pair time
    extra = None
    ms_single = None
    if rank == 0 and args.no_extras:
        extra = {"region_ms_per_step": {"median": elapsed / args.steps * 1e3, "min": min(regions) / args.steps * 1e3,
                                        "max": max(regions) / args.steps * 1e3, "regions": len(regions)},
                 "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3, "note": "--no-extras: secondary figures skipped"}
    elif rank == 0:
        reps = max(5, min(20, args.steps))
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            ctx.select_async(0, 1, True, FB_OUT1, NFEAT)      # SELECTING_ALL on the resident level-0 pyramid
        ctx.sync()
        ms_select = (time.perf_counter() - t) / reps * 1e3
        singles = []
        for _ in range(5):
            t = time.perf_counter()
            for i in range(args.steps):                        # the same K steps on ONE stream (one pair in flight)
                a = 0 if i % NSETS == 0 else 2
                ctx.build_pyramids_batch([a, a + 1])
                ctx.track_async(a, a + 1, FB_SEL, FB_OUT0 if i % 2 == 0 else FB_OUT1, NFEAT)
            ctx.sync()
            singles.append((time.perf_counter() - t) / args.steps * 1e3)
        ms_single = statistics.median(singles)
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
        extra = {"region_ms_per_step": {"median": elapsed / args.steps * 1e3, "min": min(regions) / args.steps * 1e3,
                                        "max": max(regions) / args.steps * 1e3, "regions": len(regions)},
                 "overlapped_ms_per_pair": elapsed / args.steps * 1e3,
                 "pcie_pipelined_ms_per_pair": ms_pipe, "pcie_pipelined_features_per_s": NFEAT / (ms_pipe * 1e-3),
                 "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3, "latency_ms_per_pair_synchronised": ms_latency,
                 "single_stream_ms_per_pair": ms_single, "single_stream_features_per_s": NFEAT / (ms_single * 1e-3),
                 "single_stream_runs_ms": singles,
                 "ms_per_select_5000": ms_select,
                 "pcie_inclusive_ms_per_pair": ms_pcie, "pcie_inclusive_features_per_s": NFEAT / (ms_pcie * 1e-3),
                 "note": "ms_per_frame_pair = single_stream_ms_per_pair: one pair at a time on ONE stream, no overlap with other "
                         "pairs (ms_per_step is the inverse throughput with pairs_in_flight pairs overlapping).  "
                         "pcie_inclusive = H2D of two u8 frames from pageable host memory + pyramids + track + D2H of the "
                         "records, synchronised per pair; pcie_pipelined = the same bytes with klt_upload_u8_async from "
                         "pinned memory on a copy stream and the records read back every 16 pairs"}

    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        cpu = cpu_baseline(p, f0, f1, fl, NFEAT, "cfg-2 (1920x1080, 5000 features)")

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = base_line(world * NFEAT * args.steps / elapsed, world, args.steps, args.warmup, ms_per_step,
                         ms_single if ms_single is not None else ms_per_step,
                         "cfg-2: one 1920x1080 synthetic pair per GPU, 5000 features, 7x7 window, "
                         "3 pyramid levels (subsampling 4), translation only; inputs resident in HBM",
                         extra_cfg={
                             "pipelining": (("none (one HIP stream)" if nctx == 1 else
                                             "groups of steps go round-robin to %d contexts, one HIP stream each, no ordering between them "
                                             "(pairs are independent)" % nctx) +
                                            ("; every step does the full work of one pair" if B == 1 else
                                             "; the %d steps (pairs) of a group share every launch of their context: one batched pyramid "
                                             "build for their %d frames, one tracker launch for their %d feature lists -- every step still "
                                             "does the full work of one pair" % (B, 2 * B, B)) +
                                            ("; a context rebuilds the same frame slots every group (the pyramid planes the tracker reads "
                                             "are still in the Infinity Cache)" if NSETS == 1 else "; alternate groups of a context use two sets of frame slots")),
                             "pairs_in_flight": nctx * B, "contexts": nctx, "pairs_per_launch": B, "slot_sets": NSETS,
                             "features_per_pair": NFEAT, "pairs_per_step": world, "tracked": tracked,
                             "recovered_shift_px": shift, "imposed_shift_px": list(synth.DEFAULT_SHIFT),
                             "rccl_ranks": world if distributed else 0,
                             "parallelism": "1 pair per GPU" + (", RCCL all-gather (libkltgpu side stream) of the [%d steps x 5000] record "
                                                                "table every %d steps" % (GATHER_EVERY, GATHER_EVERY) if distributed else "")})
        line.update(parity)
        line["roofline"], line["cpu_baseline"], line["extra"] = roofline, cpu, extra
    for cx in ctxs:
        cx.close()
    if line is not None:
        emit(json_fd, line)
        if parity and not parity.get("parity_checked") and "max_abs_dx" in parity:
            raise SystemExit("timed records differ from the oracle: %r" % parity)


if __name__ == "__main__":
    main()
