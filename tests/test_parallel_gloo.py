"""N > 1 path on CPU: two gloo processes shard frame pairs and gather the feature records."""
import os
import socket
import subprocess
import sys

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

out = track_pairs_sharded(track_pair, N_PAIRS, N_FEAT, world, rank, torch, dist, dst=0)
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


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KLT_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GLOO_OK" in outs[0]
