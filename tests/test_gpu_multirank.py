"""Several ranks of the multi-GPU path on ONE GPU: the ranks share device 0 (KLT_RANKS_SHARE_DEVICE) and librccl is replaced by
tests/stub_rccl (KLT_RCCL_LIB), a file-based stand-in with RCCL's entry points -- RCCL itself refuses two ranks of a communicator on
one device.  The stand-in is ASYNCHRONOUS (round 4): a call only enqueues, the exchange runs in stream order inside a host function, the
operations of a group are carried out in random order with random delays, receives may be posted before their sends exist.  Everything else is the real thing: bench.py's launcher and rendezvous, one process per rank, libkltgpu.so's comm.hip
(gather with per-rank counts, all-gather of the record tables, send / receive of the feature list, barrier and max over ranks), the
per-rank seeds and shard arithmetic, the cross-rank checks against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

STUB_DIR = os.path.join(REPO, "tests", "stub_rccl")
STUB = os.path.join(STUB_DIR, "libstubrccl.so")


@pytest.fixture(scope="module")
def stub_rccl():
    src = os.path.join(STUB_DIR, "stub_rccl.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", STUB,
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], check=True)
    return STUB


def run_ranks(args, tmp_path, stub, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE", "MASTER_PORT")}
    env.update(KLT_RANKS_SHARE_DEVICE="0", KLT_RCCL_LIB=stub, KLT_STUB_RCCL_DIR=str(tmp_path / "mail"), KLT_COMM_TIMEOUT_MS="240000")
    os.makedirs(env["KLT_STUB_RCCL_DIR"], exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_cfg2_ranks_with_their_own_pairs_and_the_all_gather(world, tmp_path, stub_rccl):
    """`bench.py --gpus N`: every rank runs its own distinct pairs (seeds rank * NP + 1 ...), each context's record table is all-gathered
    every step; rank 0 checks all of its own pairs against the oracle AND pair 0 of the last rank as it arrived through the gather."""
    line = run_ranks(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--repeats", "5", "--resident-pairs", "4", "--inflight", "2", "--batch", "2",
                      "--no-cpu-baseline", "--no-extras", "--min-timed-s", "0"], tmp_path, stub_rccl)
    assert line["n_gpus"] == world and line["config"]["rccl_ranks"] == world and line["config"]["pairs_per_step"] == 4 * world
    assert line["parity_checked"] is True and line["parity_cases"] == 5 and line["max_abs_dx"] <= 1e-3
    assert "as received through the all-gather" in line["parity_what"]
    assert line["value"] > 0 and line["scaling"] == "weak"
    val = line["extra"]["rccl_validation"]               # what the communicators themselves report, all-reduced
    assert val["rccl_ranks"] == val["rccl_ranks_largest_communicator"] == val["rccl_rank_ids_seen"] == val["world_size"] == world
    assert 0 < val["per_rank_ms_per_step"]["min"] <= val["per_rank_ms_per_step"]["max"] <= line["ms_per_step"] * 1.0001
    assert "rccl_validation_failed" not in val


def test_cfg2_three_contexts_per_rank_as_in_the_default_arrangement(tmp_path, stub_rccl):
    """The default line runs three contexts per rank (three record tables, three all-gathers per step through one communicator each,
    chained in issue order): the same at a small size, two ranks."""
    line = run_ranks(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "5", "--resident-pairs", "6", "--inflight", "3", "--batch", "2",
                      "--no-cpu-baseline", "--no-extras", "--min-timed-s", "0"], tmp_path, stub_rccl)
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["contexts"] == 3 and line["config"]["pairs_per_step"] == 12
    assert line["parity_checked"] is True and line["parity_cases"] == 7 and line["max_abs_dx"] <= 1e-3


def test_cfg4_shards_of_unequal_size_through_the_gather_with_counts(tmp_path, stub_rccl):
    """7 pairs over 2 ranks (4 + 3) and over 3 ranks (3 + 2 + 2): klt_gatherv_featbuf_async with a count per rank; rank 0 holds the whole
    [7 x 2000] table, its own shard equals what it produced and the batch's last pair -- tracked by another rank -- equals the
    oracle's."""
    for world, shards in ((2, [4, 3]), (3, [3, 2, 2])):
        line = run_ranks(["--config", "cfg4", "--gpus", str(world), "--pairs", "7", "--steps", "2", "--warmup", "1", "--repeats", "5",
                          "--no-cpu-baseline", "--min-timed-s", "0"], tmp_path / str(world), stub_rccl)
        cfg = line["config"]
        assert cfg["pairs_per_rank"] == shards and cfg["rccl_ranks"] == world and cfg["gathered_table_ok"] is True
        assert line["extra"]["rccl_validation"]["rccl_rank_ids_seen"] == world and "rccl_validation_failed" not in line["extra"]["rccl_validation"]
        assert line["parity_checked"] is True and line["parity_cases"] == shards[0] + 1, line      # every pair of rank 0 + the last pair as gathered
        assert "as gathered" in line["parity_what"]


def test_cfg5_blocks_hand_the_feature_list_from_rank_to_rank(tmp_path, stub_rccl):
    """`bench.py --config cfg5 --gpus 2`: ONE 4K sequence, frames 0..7 on rank 0 and 7..14 on rank 1, the feature list travelling from
    rank 0 to rank 1 as a baton (klt_sendrecv_featbuf_async) and both blocks' final lists gathered on rank 0.  The same 14 steps on a
    single context in this process give the same two lists (sha256 of x, y, val)."""
    sys.path.insert(0, REPO)
    import bench
    from pyfeaturetrack_amd import synth
    from pyfeaturetrack_amd.backend import Context
    line = run_ranks(["--config", "cfg5", "--gpus", "2", "--steps", "7", "--repeats", "5", "--min-timed-s", "0"], tmp_path, stub_rccl)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["rccl_ranks"] == 2 and len(cfg["live_after_each_block"]) == 2
    w, h, n, B = 3840, 2160, 20000, 7
    tc = bench.cfg2_context()
    tc.max_residue = 10.0
    c = Context(0)
    try:
        c.configure(tc)
        base = synth.synth_base(w, h, 4)
        c.upload(0, synth.synth_frame(w, h, 4, 0, base=base))
        c.build_pyramids(0)
        fl, placed = c.select(0, n, use_pyramid=True)
        c.featbuf_upload(0, fl)
        want = []
        for k in range(1, 2 * B + 1):
            c.upload(k % 2, synth.synth_frame(w, h, 4, k, base=base))
            c.build_pyramids(k % 2, sync=False)
            c.track_async((k - 1) % 2, k % 2, (k - 1) % 2, k % 2, n)
            c.select_async(k % 2, 2, True, k % 2, n)
            if k % B == 0:
                want.append(bench.list_digest(c.featbuf_download(k % 2, n)))
    finally:
        c.close()
    assert cfg["list_sha16_after_each_block"] == want


def test_a_rank_that_dies_takes_the_run_down(tmp_path, stub_rccl):
    """A rank that vanishes after the warm-up, with its peers in (or about to enter) a collective: the launcher sees its exit code, stops
    the others at once -- none of them is left waiting -- and bench.py exits non-zero without a JSON line."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE", "MASTER_PORT")}
    env.update(KLT_RANKS_SHARE_DEVICE="0", KLT_RCCL_LIB=stub_rccl, KLT_STUB_RCCL_DIR=str(tmp_path / "mail"), KLT_BENCH_DIE_RANK="1")
    os.makedirs(env["KLT_STUB_RCCL_DIR"], exist_ok=True)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "5",
                        "--resident-pairs", "4", "--inflight", "2", "--batch", "2", "--no-cpu-baseline", "--no-extras", "--min-timed-s", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 60, "the surviving rank was left waiting"
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_drivers_multi_gpu_launch_line(tmp_path, stub_rccl):
    """The driver's own launch form for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks on this box's one GPU: RANK / LOCAL_RANK / WORLD_SIZE come
    from the launcher, the rendezvous file is named after its process and port, rank 0 alone prints the JSON line."""
    import importlib.util
    if importlib.util.find_spec("torch") is None:          # (not imported here: torch brings its own HIP / RCCL libraries into the process)
        pytest.skip("torch.distributed.run is not installed")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE", "MASTER_PORT")}
    env.update(KLT_RANKS_SHARE_DEVICE="0", KLT_RCCL_LIB=stub_rccl, KLT_STUB_RCCL_DIR=str(tmp_path / "mail"), TMPDIR=str(tmp_path))
    os.makedirs(env["KLT_STUB_RCCL_DIR"], exist_ok=True)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--repeats", "5", "--resident-pairs", "4", "--inflight", "2", "--batch", "2", "--no-cpu-baseline", "--no-extras", "--min-timed-s", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["parity_checked"] is True and line["parity_cases"] == 5


# ------------------------------------------------------------------------------------------------ ordering stress
@pytest.mark.parametrize("world", [4, 8])
def test_ordering_stress_with_four_and_eight_ranks(world, tmp_path, stub_rccl):
    """Everything the multi-GPU path does between ranks, with `world` rank processes on the one GPU and the asynchronous stand-in shuffling
    and delaying the operations of every group: (a) the default arrangement of the headline -- three contexts per rank, each with its own
    communicator and record table, three all-gathers per step chained in issue order; (b) cfg-4 with 257 pairs (shards of 33 and 32), the
    gather with a count per rank; (c) cfg-5's baton through all ranks.  50 steps of (a) and (b), 5 passes of the baton through every rank;
    rank 0 checks its own records AND what the last rank contributed against the oracle every time."""
    steps = ["--steps", "10", "--warmup", "1", "--repeats", "5", "--min-timed-s", "0", "--no-cpu-baseline"]
    a = run_ranks(["--gpus", str(world), "--resident-pairs", "6", "--inflight", "3", "--batch", "2", "--no-extras"] + steps, tmp_path / "a", stub_rccl)
    assert a["n_gpus"] == world and a["config"]["rccl_ranks"] == world and a["config"]["contexts"] == 3
    assert a["config"]["pairs_per_step"] == 6 * world and a["parity_checked"] is True and a["parity_cases"] == 7 and a["max_abs_dx"] <= 1e-3
    b = run_ranks(["--config", "cfg4", "--gpus", str(world), "--pairs", "257"] + steps, tmp_path / "b", stub_rccl)
    shards = b["config"]["pairs_per_rank"]
    assert sum(shards) == 257 and max(shards) - min(shards) == 1 and len(shards) == world
    assert b["config"]["gathered_table_ok"] is True and b["parity_checked"] is True and b["parity_cases"] == shards[0] + 1
    c = run_ranks(["--config", "cfg5", "--gpus", str(world), "--steps", "7", "--repeats", "5", "--min-timed-s", "0", "--no-cpu-baseline"],
                  tmp_path / "c", stub_rccl)
    cfg = c["config"]
    assert c["n_gpus"] == world and cfg["rccl_ranks"] == world and cfg["live_after_each_block"] == [20000] * world
    assert len(set(cfg["list_sha16_after_each_block"])) == world


def test_a_timed_out_collective_does_not_hang_the_teardown(tmp_path, stub_rccl):
    """after klt_comm_wait has given up (KLT_ERR_TIMEOUT: the peer is gone), the context must still close -- the communicator
    is poisoned, nothing waits for its stream again, klt_destroy returns at once -- and the rank leaves non-zero.  Two ranks on the one
    GPU; rank 1 joins the communicator and exits without taking part in the gather."""
    import time
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from pyfeaturetrack_amd import parallel
from pyfeaturetrack_amd._abi import KltCommTimeout, exit_on_comm_timeout
from pyfeaturetrack_amd.backend import Context, FEAT_DTYPE
rank, local, world = parallel.world_from_env()
ctx = Context(0)
parallel.init_communicators([ctx], rank, world)
if rank == 1:
    os._exit(0)                                            # gone before the collective
ctx.comm_set_timeout(1500.0)
ctx.featbuf_upload(1, np.zeros(100, FEAT_DTYPE))
ctx.gather_featbuf_async(1, 2, 100, 0)
t = time.monotonic()
try:
    ctx.comm_wait()
    print("NO TIMEOUT")
    os._exit(1)
except KltCommTimeout as e:
    waited = time.monotonic() - t
    try:
        ctx.comm_wait()                                    # poisoned: fails at once
        os._exit(1)
    except KltCommTimeout:
        pass
    t = time.monotonic()
    ctx.close()                                            # must not wait for the dead collective
    print("TIMEOUT after %%.1f s, close took %%.2f s" %% (waited, time.monotonic() - t), flush=True)
    exit_on_comm_timeout(e)
''' % REPO)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE", "MASTER_PORT")}
    env.update(KLT_RANKS_SHARE_DEVICE="0", KLT_RCCL_LIB=stub_rccl, KLT_STUB_RCCL_DIR=str(tmp_path / "mail"), WORLD_SIZE="2",
               KLT_RDZV_FILE=str(tmp_path / "ids"))
    os.makedirs(env["KLT_STUB_RCCL_DIR"], exist_ok=True)
    t0 = time.monotonic()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    out0, err0 = procs[0].communicate(timeout=120)
    procs[1].wait(timeout=60)
    assert procs[0].returncode == 3, (procs[0].returncode, out0, err0[-2000:])
    assert "TIMEOUT after" in out0 and "exiting with code 3" in err0
    took = float(out0.split("close took")[1].split()[0])
    assert took < 5.0 and time.monotonic() - t0 < 90
