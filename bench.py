#!/usr/bin/env python3
"""Headline benchmark: features tracked / s and ms per frame pair (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

Workload at every N = BASELINE cfg-2 (1920x1080 synthetic pairs, 5000 features each, 7x7 window, 3 pyramid levels / subsampling 4,
translation only).  `--resident-pairs` (default 72) DISTINCT pairs per GPU (seeds rank * 72 + 1 ...) are resident in HBM with their
own frame slots, pyramids (4.2 GB per GPU) and feature lists; a *step* is one pass of the hot path over that batch: for every pair,
build the image / gradx / grady pyramids of both frames from the u8 frames in HBM, then track every live feature coarse-to-fine
(device-resident feature records in, device-resident records out) -- 72 KLTTrackFeatures-equivalents per step.  The timed region
therefore streams 4.2 GB of distinct frames and pyramids per step: nothing it reads was left in the 256 MB Infinity Cache by the
step before (round 2 rebuilt the same four slots from the same two frames, `extra.cache_resident_ms_per_pair` keeps that figure).
With N > 1 every rank runs its own 72 pairs (weak scaling: the path shards by frame pair with no data-path exchange) and the
16-byte records of a step are collected in one device-side [pairs x features] table per context and gathered to every rank with
one RCCL all-gather per table, issued by libkltgpu.so on its side stream (event-ordered behind the last tracker launch of the step,
overlapped with the next step's kernels).  No torch anywhere.

Launching.  `--gpus N` with N > 1 and no RANK in the environment: this process -- before it touches the GPU in any way -- starts N
fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous file in their environment), forwards rank 0's JSON line and
exits non-zero if any rank failed.  Under an external launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus
N ...`) the same variables are already set and every process is a rank.  The RCCL unique id travels through the rendezvous file
(pyfeaturetrack_amd/parallel.py).

Consecutive groups of `--batch` pairs (default 8) go round-robin to `--inflight` contexts (default 3; one HIP stream each, nothing
ordering them): frame pairs are independent, so the pairs of a group share every launch of their context (one batched pyramid build,
one tracker launch -- the reference's workload for a stereo rig or two cameras) and the GPU overlaps the kernels of different groups.
Every pair gets the full work of one KLTTrackFeatures call.  `ms_per_frame_pair` (= `extra.single_stream_ms_per_pair`) is one pair
at a time on one stream, rotating through the resident pairs.  The K-step timed region (barrier + synchronise on both sides, MAX
over ranks) is repeated until at least `--repeats` regions AND 6 s of timed work are in (never fewer than 5 regions); `ms_per_step`
is the median region, the spread is in `extra.region_ms_per_step`.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  parity_checked -- the records of the last timed step of EVERY resident pair equal the CPU oracle's (the run fails otherwise);
  roofline     -- the dominant kernel of the step (largest share of device time), timed by its dispatches' own start / stop
                  timestamps in a second pass over the same pairs (events on every launch would distort the un-instrumented
                  `value`), next to the whole step (`step_frac`) and the per-kernel table; bench.py refuses to print a line in
                  which any fraction of peak exceeds 1;
  cpu_baseline -- the CPU oracle (oracle/klt_oracle.c, a bit-exact port of the reference's Python/Cython/SciPy path) on the same
                  workload, 1 thread, rank 0, N = 1 only.
`--config cfg1|cfg3|cfg4|cfg5` run the other BASELINE configs the same way: timed regions, records checked against the oracle,
`roofline` with the per-kernel table, `cpu_baseline`.  The default run (cfg-2, one GPU) runs all four at their BASELINE counts as child
processes of its own after the headline and keeps a compact record of each in `extra.configs` (benchlib/sweep.py); a config whose
records differ from the oracle's makes the whole run exit non-zero.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the parts (benchlib/): shared timing / roofline / oracle helpers and one module per BASELINE config.  Importing them does not touch
# the GPU (the library is bound on first use).  The helpers stay importable as `bench.<name>` (tests use a few of them).
from benchlib import common                                              # noqa: E402
from benchlib.common import *                                            # noqa: E402,F401,F403
from benchlib.cfg1 import run_cfg1                                       # noqa: E402
from benchlib.cfg2 import link_rates, run_cfg2, sequence_from_host      # noqa: E402,F401
from benchlib.api_figures import api_figures                            # noqa: E402,F401
from benchlib.cfg3 import affine_bytes, cfg3_context, cfg3_frames, run_cfg3        # noqa: E402,F401
from benchlib.cfg4 import run_cfg4                                       # noqa: E402
from benchlib.cfg5 import run_cfg5, run_cfg5_blocks                      # noqa: E402,F401
from benchlib.dryrun import dry_run                                      # noqa: E402


# ============================================================================================ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=25,
                    help="the K-step timed region is run at least this often AND until --min-timed-s of timed work are in (median reported; "
                         "fewer, never below 5, when a region is long)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed hot-path work before the W warm-up steps: the GPU needs ~10 ms of load to reach its steady clocks / "
                         "cache state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures (selection, one pair at a time, PCIe-inclusive, Python API): a profiler then sees only "
                         "the launches of the timed regions and of the roofline pass, all of the headline's size")
    ap.add_argument("--no-api", action="store_true", help="skip the reference-shaped Python API figures in `extra`")
    ap.add_argument("--no-config-sweep", action="store_true",
                    help="skip extra.configs (the default cfg-2 run otherwise runs --config cfg1, cfg3, cfg4 --pairs 256 and cfg5 --frames 512, each in a "
                         "child process of its own, after the headline, and folds a compact record of each into its line); --no-extras implies it")
    ap.add_argument("--no-sequences", action="store_true", help="skip extra.sequence_from_host (a 1080p and a 4K sequence fed from pinned host memory)")
    ap.add_argument("--config", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the headline line); cfg4 = the 256-pair batch sharded over --gpus ranks; the others are "
                         "the remaining BASELINE configs on one GPU")
    ap.add_argument("--pairs", type=int, default=256, help="total pairs per step for --config cfg4 (sharded over the ranks)")
    ap.add_argument("--frames", type=int, default=512, help="length of the sequence of --config cfg5 on one GPU (BASELINE: 512 frames); a timed region is whole passes over it")
    ap.add_argument("--resident-pairs", type=int, default=72,
                    help="cfg2: distinct synthetic pairs resident per GPU, all of them processed by every step (72 pairs = 4.2 GB of frames "
                         "and pyramids: a step's working set is 16x the 256 MB Infinity Cache); a multiple of --inflight x --batch")
    ap.add_argument("--inflight", type=int, default=3,
                    help="contexts per GPU (one HIP stream each, no events between them): consecutive groups of --batch pairs go "
                         "round-robin to them, so kernels of different groups overlap; 1 = a single stream.  With eight pairs per launch "
                         "3 contexts read 0.6-2.5 %% above 2 in every session (163.3 / 159.3, 158-160 / 154-155, 153.0 / 152.1 M features/s); "
                         "4 x 4 pairs 7 %% below")
    ap.add_argument("--batch", type=int, default=8, choices=[1, 2, 4, 8, 16],
                    help="pairs that share every launch of a context: one batched pyramid build for their frames and one tracker launch "
                         "for their feature lists.  On distinct resident pairs (nothing to keep in a cache) longer launches win: 2 contexts x "
                         "1 / 2 / 4 / 8 / 16 pairs per launch read 0.0414 / 0.0358 / 0.0345 / 0.0332 / 0.0333 ms per pair (the level-0 kernel pays "
                         "its ramp and tail once per launch: 11.7 us per frame in a 4-frame launch, 9.8 in a 16-frame launch)")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S,
                    help="the timed regions of a run add up to at least this many seconds (0 for profiler passes, which replay every kernel)")
    args = ap.parse_args()
    common.MIN_TIMED_S = args.min_timed_s

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
    return {"cfg1": run_cfg1, "cfg2": run_cfg2, "cfg3": run_cfg3, "cfg4": run_cfg4, "cfg5": run_cfg5}[args.config](args, json_fd)


if __name__ == "__main__":
    from pyfeaturetrack_amd._abi import KltCommTimeout, exit_on_comm_timeout
    try:
        main()
    except KltCommTimeout as e:                 # a peer is gone: report and leave non-zero, without waiting for anything
        exit_on_comm_timeout(e)
