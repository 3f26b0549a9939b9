"""KLT_BENCH_DRYRUN=1: the launcher, rendezvous and line plumbing of bench.py without a GPU (CPU tests)."""
from .common import *            # noqa: F401,F403 -- the shared helpers, constants and the modules they import (np, os, time, ...)


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
                   "gatherv_counts": [len(s["pairs"]) for s in seen],
                   "local_ranks": [s["local_rank"] for s in seen], "spawned": os.environ.get("KLT_SPAWNED") == "1"})

