"""What the configs of bench.py share: the constants of the line, algorithmic-byte bookkeeping, the checker (the CPU oracle) and the
CPU baseline, rank plumbing and timed regions, per-kernel tables and the `roofline` object, the line's sanity checks."""
import argparse
import hashlib
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# importing the package does not touch the GPU (the library is bound on first use)
from pyfeaturetrack_amd import parallel, synth                          # noqa: E402
from pyfeaturetrack_amd.backend import Context                          # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                  # noqa: E402
from pyfeaturetrack_amd.params import affine_params_from_tc, params_from_tc   # noqa: E402

WIDTH, HEIGHT, NFEAT = 1920, 1080, 5000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TOL_PX = 1e-3             # north_star: sub-pixel x/y within 1e-3 (observed: 0)
DTYPE = "f32 (convolutions accumulate in f64)"
MIN_TIMED_S = 6.0         # the timed regions of a run add up to at least this much GPU work (the driver samples the GPU every 5 s:
                          # r03 saw 0 of 4 samples busy with 2 s of timed work inside an 18 s run)
HBM_ACHIEVABLE_GBS = 6300.0   # MI355X_MICROARCH.md: what a streaming kernel sustains of the 8 TB/s
ORACLE_NOTE = "oracle/klt_oracle.c (pinned to reference-generated goldens)"


def cfg2_context():
    tc = KLT_TrackingContext()
    tc.nPyramidLevels = 3
    tc.subsampling = 4
    tc.KLTUpdateTCBorder()          # border 120
    return tc


def level_pixels(p, ncols, nrows):
    n, dims = [], (ncols, nrows)
    for _ in range(p.nPyramidLevels):
        n.append(dims[0] * dims[1])
        dims = (dims[0] // p.subsampling, dims[1] // p.subsampling)
    return n


def pyramid_bytes(p, ncols, nrows, b_in=1):
    """N0 (b_in + 4) + sum 4 (N_{l-1} + N_l) + sum 12 N_l -- one frame"""
    n = level_pixels(p, ncols, nrows)
    return n[0] * (b_in + 4) + sum(4 * (n[l - 1] + n[l]) for l in range(1, len(n))) + sum(12 * v for v in n)


def track_bytes(p, stats, nfeat):
    """sum over features and levels of 12 (w+1)(h+1) (1 + iterations) + 24 per record; `stats` = totals of the
    device counters (klt_track_stats) over the launches they cover, nfeat = records those launches read and wrote"""
    L = p.nPyramidLevels
    foot = 12.0 * (p.window_width + 1) * (p.window_height + 1)
    return foot * (sum(stats["level_visits"][:L]) + sum(stats["iterations"][:L])) + 24.0 * nfeat


def algorithmic_bytes(p, ncols, nrows, stats, nfeat):
    """per pair (pyramid bytes of one frame, tracker bytes); `stats` = per-pair averages"""
    return pyramid_bytes(p, ncols, nrows), track_bytes(p, stats, nfeat)


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


# ============================================================================= the checker (oracle) and the CPU baseline
def load_oracle():
    """oracle/klt_oracle.py -- the checker and the cpu_baseline leg only; never the thing measured.  None if it is not built."""
    try:
        from oracle import klt_oracle as ko
        ko.lib()
        return ko
    except (ImportError, OSError) as e:
        print("oracle unavailable: %s" % e, file=sys.stderr)
        return None


def oracle_track(ko, p, f0, f1, fl, threads=1):
    """The CPU oracle's records for one pair."""
    a0, a1 = f0.astype(np.float32), f1.astype(np.float32)
    ofl = fl.copy()
    ko.set_threads(threads)
    ko.track_features(p, ko.Pyramids(p, a0), ko.Pyramids(p, a1), ofl)
    ko.set_threads(1)
    return ofl


def records_equal(out, ofl):
    """(status codes equal, max |dx, dy|) of records `out` against the oracle's `ofl`"""
    same_val = bool(np.array_equal(out["val"], ofl["val"]))
    dx = float(max(np.abs(out["x"].astype(np.float64) - ofl["x"]).max(), np.abs(out["y"].astype(np.float64) - ofl["y"]).max())) if len(out) else 0.0
    return same_val, dx


def parity_summary(checks, what):
    """{parity_checked, ...} from [(label, status codes equal, max |dx|)]; an empty list = unchecked"""
    if not checks:
        return {"parity_checked": False, "parity_note": "oracle library not built on this box"}
    worst = max(c[2] for c in checks)
    same = all(c[1] for c in checks)
    bad = [c[0] for c in checks if not c[1] or c[2] > TOL_PX]
    out = {"parity_checked": bool(same and worst <= TOL_PX), "max_abs_dx": worst, "status_codes_equal": same,
           "parity_tolerance_px": TOL_PX, "parity_against": ORACLE_NOTE, "parity_cases": len(checks), "parity_what": what}
    if bad:
        out["parity_failed_cases"] = bad[:8]
    return out


def fail_on_parity(par):
    if par and not par.get("parity_checked") and "max_abs_dx" in par:
        raise SystemExit("timed records differ from the oracle: %r" % par)


def cpu_time(fn, budget_s=10.0, max_reps=200):
    """(seconds per call, calls): one call to size the sample, then about `budget_s` of them"""
    t = time.perf_counter()
    fn()
    t1 = time.perf_counter() - t
    reps = int(max(2, min(max_reps, budget_s / max(t1, 1e-4))))
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps, reps


def cpu_baseline_of(ko, one_step, nfeat, what, all_cores=True, budget_s=10.0, reference_python_survey=None):
    """Oracle timed on the host: a bounded sample of the same workload (about 10-20 s of CPU work).  `one_step()` = one step of the
    config on the CPU; value = nfeat / seconds."""
    if ko is None:
        return None
    ko.set_threads(1)
    dt, reps = cpu_time(one_step, budget_s)
    out = {"value": nfeat / dt, "unit": "features/s", "cores": 1, "kind": "port", "ms_per_step": dt * 1e3,
           "sample": "%d x (%s), oracle/klt_oracle.c, 1 thread" % (reps, what)}
    if reference_python_survey:
        # the reference ITSELF (Python / Cython / SciPy; it cannot travel to the GPU box) as the survey timed
        # it in its own container, next to the port -- context, not a measurement of this run
        ref = dict(reference_python_survey)
        ref["port_over_reference"] = (nfeat / dt) / ref["features_per_s"]
        ref["note"] = ("TimSC/PyFeatureTrack itself on this workload, measured by the survey (BASELINE.md section 2: Intel Xeon @ 2.10 GHz, "
                       "one thread; best of 3); port_over_reference = this run's one-thread oracle / that figure -- two different hosts")
        out["reference_python_survey"] = ref
    if all_cores:
        # the same port on the host cores this process may actually use (OpenMP over image lines / features; bit-identical
        # results).  Time-bounded: a container with a CPU quota can make many threads slower than one.
        ncores = ko.set_threads(usable_cores())
        t = time.perf_counter()
        reps_all = 0
        while reps_all < 40 and (reps_all < 2 or time.perf_counter() - t < 4.0) and time.perf_counter() - t < 12.0:
            one_step()
            reps_all += 1
        dt_all = (time.perf_counter() - t) / reps_all
        ko.set_threads(1)
        out["all_cores"] = {"value": nfeat / dt_all, "cores": ncores, "ms_per_step": dt_all * 1e3,
                            "sample": "%d x the same step, OpenMP over image lines and features" % reps_all}
    return out


def list_digest(fl):
    """sha256 (16 hex digits) of the (x, y, val) columns of a feature list"""
    cols = np.stack([fl["x"].view(np.int32), fl["y"].view(np.int32), fl["val"].astype(np.int32)], axis=1)
    return hashlib.sha256(np.ascontiguousarray(cols).tobytes()).hexdigest()[:16]


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
        self.gpus_asked = args.gpus
        self.local_s = []                                  # this rank's own elapsed time of every timed region (the line's is the MAX over ranks)

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
        self.local_s.append(el)
        return self.max_over_ranks(el), enq

    def validation(self, steps_per_region):
        """The N > 1 line validates itself.  What RCCL reports -- klt_comm_info of every context of every rank, all-reduced:
        `rccl_ranks` = the SMALLEST communicator any rank is in, `rccl_rank_ids_seen` = how many different rank numbers answered -- and every
        rank's own median region time, min / max over the ranks (the line's ms_per_step is the max by contract).  Collective: every rank
        calls it at the same point.  {} without communicators."""
        if not self.distributed:
            return {}
        info = [cx.comm_info() for cx in self.ctxs]
        sizes, ids = [i[0] for i in info], {i[1] for i in info}
        onehot = [1.0 if r in ids else 0.0 for r in range(12)]
        loc = sorted(self.local_s)[len(self.local_s) // 2] / max(1, steps_per_region) * 1e3 if self.local_s else 0.0
        v = self.ctxs[0].comm_allreduce_max([-float(min(sizes)), float(max(sizes)), loc, -loc] + onehot)
        out = {"rccl_ranks": int(-v[0]), "rccl_ranks_largest_communicator": int(v[1]), "rccl_rank_ids_seen": int(sum(v[4:])) if self.world <= 12 else None,
               "gpus_asked": self.gpus_asked, "world_size": self.world,
               "per_rank_ms_per_step": {"max": v[2], "min": -v[3], "note": "every rank's own median timed region / steps; the line's ms_per_step is the MAX over ranks of each region"}}
        bad = []
        if out["rccl_ranks"] < self.gpus_asked or out["rccl_ranks"] != self.world or out["rccl_ranks_largest_communicator"] != self.world:
            bad.append("communicators of %d..%d ranks, WORLD_SIZE %d, --gpus %d" % (out["rccl_ranks"], out["rccl_ranks_largest_communicator"], self.world, self.gpus_asked))
        if out["rccl_rank_ids_seen"] is not None and out["rccl_rank_ids_seen"] != self.world:
            bad.append("%d different rank numbers answered, %d expected" % (out["rccl_rank_ids_seen"], self.world))
        if bad:
            out["rccl_validation_failed"] = "; ".join(bad)
        return out

    @staticmethod
    def fail_on_validation(val):
        """after the line is out: a run in which some rank saw fewer peers than --gpus is not a measurement of --gpus GPUs"""
        if val.get("rccl_validation_failed"):
            raise SystemExit("the ranks of this run do not add up: " + val["rccl_validation_failed"])


class OneGpu:
    """The same bracket for the single-GPU configs (no communicator)."""
    distributed = False
    rank, world = 0, 1

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)

    def max_over_ranks(self, v):
        return v

    def timed(self, fn):
        for cx in self.ctxs:
            cx.sync()
        t0 = time.perf_counter()
        fn()
        enq = time.perf_counter() - t0
        for cx in self.ctxs:
            cx.sync()
        return time.perf_counter() - t0, enq


def timed_regions(ranks, run_region, repeats, min_total_s=None, budget_s=30.0):
    """K-step regions, each bracketed as the contract says, until `repeats` regions AND `min_total_s` of timed work are in (so that
    a sampler outside this process sees a busy GPU even when one region lasts a millisecond); fewer -- never below 5 -- when the
    regions are long.  Returns (median seconds per region, all regions, host enqueue seconds of the median region)."""
    min_total_s = MIN_TIMED_S if min_total_s is None else min_total_s
    el, enq = ranks.timed(run_region)
    regions = [(el, enq)]
    n = max(repeats, int(math.ceil(min_total_s / max(el, 1e-9))))
    if el * n > budget_s:
        n = max(5, int(budget_s / max(el, 1e-9)))
    n = int(ranks.max_over_ranks(n)) if ranks.distributed else n      # every rank runs the same number of regions
    while len(regions) < n:
        regions.append(ranks.timed(run_region))
    regions.sort()
    med = regions[len(regions) // 2]
    return med[0], [r[0] for r in regions], med[1]


def region_stats(regions, units, elapsed):
    """spread of the timed regions in ms per unit (`units` per region)"""
    return {"median": elapsed / units * 1e3, "min": min(regions) / units * 1e3, "max": max(regions) / units * 1e3,
            "regions": len(regions), "timed_s_total": sum(regions)}


# ------------------------------------------------------------------------------------------------- roofline bookkeeping
def timed_pass(ctx, run, mode):
    """`run()` with every launch timed: mode 1 = an event pair around each launch (it also holds the boundary to the launch before,
    ~2.6 us); mode 2 = the kernels that are one launch per call by their dispatch's own start / stop timestamps -- what rocprofv3
    reports as the kernel's duration.  {family: {launches, total_ms, bytes}}"""
    ctx.sync()
    ctx.timing_enable(mode)
    run()
    res = ctx.timing_read()
    ctx.timing_enable(False)
    return {k["name"]: k for k in res}


KLT_OPT_TRACK_TREE_SUMS = 18
# VGPRs of the two forms of the quad tracker kernels, from the compiler's metadata (tools/kernel_regs.py prints them from a fresh
# compile of track_kernels.hip; a CPU test compares)
TRACKER_VGPRS = {7: {"exact": 124, "tree": 94}, 15: {"exact": 96, "tree": 96}}


def tree_sums_probe(ctx, launch, read, bytes_per_launch, window, reps=6):
    """what bit-identity costs the tracker.  The same tracker launches (resident pyramids, the same input lists) with the
    sums added in the reference's order (the default, LDS product arrays + five serial chains) and with KLT_OPT_TRACK_TREE_SUMS (butterfly
    sums in registers: same precision, other order of the additions): duration per launch by the dispatches' timestamps, and how the
    records differ -- per call on identical inputs, not chained."""
    def one(opt):
        ctx.set_option(KLT_OPT_TRACK_TREE_SUMS, opt)
        launch()
        ctx.sync()
        r = timed_pass(ctx, lambda: [launch() for _ in range(reps)], 2).get("track")
        if not r or not r["launches"]:
            r = timed_pass(ctx, lambda: [launch() for _ in range(reps)], 1)["track"]
        launch()
        return 1e3 * r["total_ms"] / r["launches"], read().copy()
    try:
        us_exact, rec_exact = one(0)
        us_tree, rec_tree = one(1)
        us_exact2, _ = one(0)
    finally:
        ctx.set_option(KLT_OPT_TRACK_TREE_SUMS, 0)
    us_exact = min(us_exact, us_exact2)
    both = (rec_exact["val"] == 0) & (rec_tree["val"] == 0)
    flips = int((rec_exact["val"] != rec_tree["val"]).sum())
    dx = float(max(np.abs(rec_exact["x"][both] - rec_tree["x"][both]).max(), np.abs(rec_exact["y"][both] - rec_tree["y"][both]).max())) if both.any() else 0.0
    out = {"us_per_launch": us_tree, "us_per_launch_exact": us_exact, "speedup": us_exact / us_tree,
           "frac": bytes_per_launch / (us_tree * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_exact": bytes_per_launch / (us_exact * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "vgprs": TRACKER_VGPRS.get(window, {}).get("tree"), "vgprs_exact": TRACKER_VGPRS.get(window, {}).get("exact"),
           "max_abs_dx": dx, "status_flips": flips, "features": int(rec_exact.size),
           "differing_positions": int(((rec_exact["x"] != rec_tree["x"]) | (rec_exact["y"] != rec_tree["y"]))[both].sum()),
           "note": "opt-in KLT_OPT_TRACK_TREE_SUMS (off in every other figure of this line): the five window sums and the residue by a DPP butterfly "
                   "in registers instead of LDS product arrays added in the reference's sequential order"}
    return out


def kernel_table(stamped, paired, nsteps, bytes_override=None, peak=HBM_PEAK_GBS):
    """per-kernel figures of a config's step: duration per launch (dispatch timestamps where the family has them, else the event
    pair), launches per step, algorithmic bytes per launch (the library books SURVEY 8(d)'s figure per launch; the tracker's and the
    affine check's come from the device counters: `bytes_override` = {family: total bytes over the pass}), GB/s and fraction of peak"""
    out = {}
    for name, k in sorted(paired.items(), key=lambda kv: -kv[1]["total_ms"]):
        s = stamped.get(name) if stamped else None
        src = s if s and s["launches"] == k["launches"] else k
        total_bytes = (bytes_override or {}).get(name, k["bytes"])
        us = 1e3 * src["total_ms"] / src["launches"]
        gbps = total_bytes / max(src["total_ms"], 1e-9) / 1e6
        out[name] = {"us_per_launch": us, "launches_per_step": k["launches"] / nsteps,
                     "timed_by": "dispatch timestamps" if src is s else "event pair",
                     "us_per_launch_event_pair": 1e3 * k["total_ms"] / k["launches"],
                     "algorithmic_bytes_per_launch": total_bytes / k["launches"], "GBps": gbps, "frac": gbps / peak}
    return out


def roofline_of(table, nsteps, ms_per_step, peak=HBM_PEAK_GBS, dominant=None, extra=None):
    """the `roofline` object: the dominant kernel (largest share of device time) against the HBM roof, the whole step next to it"""
    dom = dominant or max(table, key=lambda n: table[n]["us_per_launch"] * table[n]["launches_per_step"])
    d = table[dom]
    step_bytes = sum(k["algorithmic_bytes_per_launch"] * k["launches_per_step"] for k in table.values())
    dev_ms = sum(k["us_per_launch"] * k["launches_per_step"] for k in table.values()) * 1e-3
    r = {"bound": "hbm", "kernel": dom, "achieved": d["GBps"], "peak": peak, "unit": "GB/s", "frac": d["frac"],
         "frac_vs_achievable": min(1.0, d["GBps"] / HBM_ACHIEVABLE_GBS), "achievable": HBM_ACHIEVABLE_GBS,
         "achievable_note": "the guide's measured streaming ceiling (6.3 TB/s of the 8 TB/s specification); frac stays against the specification",
         "traffic": None,
         "launch_us": d["us_per_launch"], "launch_us_source": d["timed_by"], "launch_us_event_pair": d["us_per_launch_event_pair"],
         "launches_per_step": d["launches_per_step"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
         "step_algorithmic_bytes": step_bytes, "step_kernel_ms": dev_ms,
         "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / peak,
         "step_frac_note": "step_algorithmic_bytes / ms_per_step / peak (the un-instrumented timed regions); step_kernel_ms = sum of the kernels' own durations",
         "kernels": table}
    r.update(extra or {})
    return r


def check_fractions(obj, path="line"):
    """Every fraction of peak in the line must be <= 1 (and every GB/s <= the peak next to it): a figure above the roof is a
    bookkeeping error (round 2 shipped step_frac 2.6 from counters that included warm-up launches), never a result."""
    bad = []

    def walk(o, p, peak):
        if isinstance(o, dict):
            peak = o.get("peak", peak) if isinstance(o.get("peak"), (int, float)) else peak
            for k, v in o.items():
                if isinstance(v, bool) or v is None:
                    continue
                if isinstance(v, (int, float)):
                    if (k == "frac" or k.endswith("_frac") or k.startswith("frac_")) and not (0.0 <= v <= 1.0):
                        bad.append("%s.%s = %r" % (p, k, v))
                    if k == "GBps" and peak and v > peak:
                        bad.append("%s.%s = %r > peak %r" % (p, k, v, peak))
                else:
                    walk(v, p + "." + k, peak)
        elif isinstance(o, list):
            for i, v in enumerate(o):
                walk(v, "%s[%d]" % (p, i), peak)

    walk(obj, path, HBM_PEAK_GBS)
    return bad


def emit(json_fd, line):
    bad = check_fractions(line)
    if bad:
        raise SystemExit("refusing to print a line with figures above the roof: " + "; ".join(bad))
    os.write(json_fd, (json.dumps(line) + "\n").encode())      # the ONE JSON line on the real stdout


def base_line(value, n_gpus, steps, warmup, ms_step, ms_pair, workload, scaling="weak", extra_cfg=None):
    cfg = {"workload": workload}
    cfg.update(extra_cfg or {})
    return {"metric": "features tracked/sec", "value": value, "unit": "features/s", "n_gpus": n_gpus, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_step, "ms_per_frame_pair": ms_pair, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": cfg, "roofline": None, "cpu_baseline": None}


def sane_iterations(stats, nfeat_total, levels, what):
    """The Newton-iteration counters must describe exactly the launches they are divided by: per feature and level between 1 and
    max_iterations (10) on average.  (Round 2 reset them before a warm-up loop.)"""
    for l in range(levels):
        per = stats["iterations"][l] / max(1, nfeat_total)
        if not (0.5 <= per <= 10.0):
            raise SystemExit("%s: %.2f Newton iterations per feature at level %d -- the counters cover other launches than the ones "
                             "they are booked on" % (what, per, l))

