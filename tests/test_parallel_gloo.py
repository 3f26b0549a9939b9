"""N > 1 path on CPU: shard arithmetic, the launcher + rendezvous of bench.py with two processes (collective
stubbed: no GPU here), and two gloo processes that shard frame pairs and gather the feature records."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

from conftest import REPO

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["KLT_REPO"])
from pyfeaturetrack_amd.backend import FEAT_DTYPE
from pyfeaturetrack_amd.parallel import shard_range, track_pairs_sharded

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N_PAIRS, N_FEAT = 7, 33

def track_pair(i):                      # deterministic stand-in for the per-pair GPU work
    r = np.zeros(N_FEAT, FEAT_DTYPE)
    r["x"] = np.arange(N_FEAT) + 1000 * i
    r["y"] = -np.arange(N_FEAT) * 0.5
    r["val"] = np.where(np.arange(N_FEAT) % 5 == 0, -4, 0)
    r["aux"] = rank
    return r

def gather(local, dst=0):               # host-level stand-in for klt_gather_featbuf_async (ranks may own unequal shards)
    counts = [None] * world
    dist.all_gather_object(counts, int(local.shape[0]))
    width = max(counts)
    pad = np.zeros((width, N_FEAT), FEAT_DTYPE)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad.view(np.int32).reshape(width, N_FEAT, 4))
    outs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, outs, dst=dst)
    if outs is None:
        return None
    return np.concatenate([o.numpy().reshape(width, N_FEAT * 4).view(FEAT_DTYPE).reshape(width, N_FEAT)[:c]
                           for o, c in zip(outs, counts)], axis=0)

out = track_pairs_sharded(track_pair, N_PAIRS, N_FEAT, world, rank, gather)
if rank == 0:
    assert out.shape == (N_PAIRS, N_FEAT), out.shape
    for i in range(N_PAIRS):
        ref = track_pair(i)
        assert np.array_equal(out[i]["x"], ref["x"]) and np.array_equal(out[i]["val"], ref["val"])
    owners = [int(out[i]["aux"][0]) for i in range(N_PAIRS)]
    assert owners == [0, 0, 0, 0, 1, 1, 1], owners
    assert list(shard_range(7, 2, 0)) == [0, 1, 2, 3] and list(shard_range(7, 2, 1)) == [4, 5, 6]
    print("GLOO_OK")
else:
    assert out is None
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_everything():
    from pyfeaturetrack_amd.parallel import shard_range
    for n in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(n, world, r)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert [len(shard_range(256, 8, r)) for r in range(8)] == [32] * 8       # BASELINE cfg-4: 32 pairs per GPU


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KLT_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GLOO_OK" in outs[0]


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE", "MASTER_PORT")}
    env.update(KLT_BENCH_DRYRUN="1", **kw)
    return env


def test_bench_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes (before touching any GPU), they meet
    through the rendezvous file, agree on the communicator ids and cover the pairs; rank 0's ONE JSON line comes back."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "cfg4", "--pairs", "7"],
                       env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line == {"dryrun": True, "n_gpus": 2, "ids_agree": True, "pairs_covered": True, "gatherv_counts": [4, 3], "local_ranks": [0, 1],
                    "spawned": True}


def test_bench_launcher_with_shards_of_unequal_size():
    """7 and 257 pairs over 2 ranks (and 3 pairs over 4: one rank owns nothing): every pair is covered exactly once."""
    for gpus, pairs in ((2, 7), (2, 257), (4, 3)):
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--config", "cfg4", "--pairs", str(pairs)],
                           env=_clean_env(), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
        assert line["pairs_covered"] and line["ids_agree"] and line["n_gpus"] == gpus, line
        assert line["gatherv_counts"] == [len(range(*divmod_range(pairs, gpus, k))) for k in range(gpus)], line


def divmod_range(n, world, rank):
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies takes the run down with a non-zero exit code instead of leaving the others in a collective."""
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "cfg4", "--pairs", "8"],
                       env=_clean_env(KLT_DRYRUN_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 45, "the surviving rank was not stopped"
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_an_external_launcher(tmp_path):
    """torch.distributed.run style: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT come from the launcher; the rendezvous file is
    named by the common parent process and the port."""
    env = _clean_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", TMPDIR=str(tmp_path))
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "cfg4", "--pairs", "6"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    line = json.loads(outs[0][0].strip())
    assert line["n_gpus"] == 2 and line["ids_agree"] and line["pairs_covered"] and line["spawned"] is False
    assert outs[1][0].strip() == ""                      # only rank 0 prints


def test_exchange_ids_is_atomic_for_late_readers(tmp_path):
    from pyfeaturetrack_amd.parallel import KLT_COMM_ID_BYTES, exchange_ids
    path = str(tmp_path / "ids")
    ids0 = exchange_ids(0, 2, 2, lambda: os.urandom(KLT_COMM_ID_BYTES), path=path)
    ids1 = exchange_ids(1, 2, 2, None, path=path, timeout=5)
    assert ids0 == ids1 and len(ids0) == 2 and all(len(i) == KLT_COMM_ID_BYTES for i in ids0)
    try:
        exchange_ids(1, 2, 3, None, path=path, timeout=0.2)          # wrong size never matches: times out instead of mis-reading
        assert False
    except TimeoutError:
        pass
