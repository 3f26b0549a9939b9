"""Multi-GPU sharding of independent frame pairs (BASELINE cfg-4) -- one process per GPU, no torch.

The KLT path shards by frame pair: every pair is independent, so ranks never exchange pixels or
pyramids.  The only communication is the gather of the 16-byte feature records at the end of a
shard (n_feat * 16 B per pair), done by libkltgpu.so itself with RCCL over xGMI
(`klt_comm_init_rank`, `klt_gather_featbuf_async`, include/klt_gpu.h).  This module holds the
host side of that: the shard arithmetic, the rendezvous that hands rank 0's RCCL unique id to the
other ranks (a file -- the ranks of one node share a filesystem), and the launcher that starts
one process per GPU.  The reference has no counterpart (single process).
"""
import os
import subprocess
import sys
import time

import numpy as np

from .backend import FEAT_DTYPE

KLT_COMM_ID_BYTES = 128


def shard_range(n_items, world, rank):
    """Contiguous block of `n_items` owned by `rank` (blocks differ by at most one item)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


# ---------------------------------------------------------------------------------------- rendezvous
def world_from_env(env=None):
    """(rank, local_rank, world) from the variables `python -m torch.distributed.run` and `spawn_ranks` both set."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0"))), int(env.get("WORLD_SIZE", "1"))


def rendezvous_file(env=None):
    """Path all ranks of one launch agree on.  `spawn_ranks` passes KLT_RDZV_FILE; under an external launcher
    (torch.distributed.run) every rank has the same parent process and MASTER_PORT, which name the file."""
    env = os.environ if env is None else env
    if env.get("KLT_RDZV_FILE"):
        return env["KLT_RDZV_FILE"]
    tag = "%s_%s_%d" % (env.get("TORCHELASTIC_RUN_ID", "none"), env.get("MASTER_PORT", "0"), os.getppid())
    return os.path.join(env.get("TMPDIR", "/tmp"), "klt_rdzv_" + tag)


def exchange_ids(rank, world, count, make_id, path=None, timeout=300.0):
    """Rank 0 creates `count` communicator ids with `make_id()` (128 bytes each) and publishes them atomically;
    the other ranks wait for the file.  Returns the list of ids on every rank."""
    path = path or rendezvous_file()
    want = count * KLT_COMM_ID_BYTES
    if rank == 0:
        blob = b"".join(make_id() for _ in range(count))
        assert len(blob) == want
        tmp = "%s.tmp%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)          # readers see nothing or everything
    else:
        t0 = time.monotonic()
        while True:
            try:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == want:
                    break
            except FileNotFoundError:
                pass
            if time.monotonic() - t0 > timeout:
                raise TimeoutError("rank %d: no communicator ids at %s after %.0f s" % (rank, path, timeout))
            time.sleep(0.01)
    return [blob[i * KLT_COMM_ID_BYTES:(i + 1) * KLT_COMM_ID_BYTES] for i in range(count)]


def init_communicators(ctxs, rank, world, path=None):
    """One RCCL communicator per context (contexts = independent HIP streams of this rank; collectives of
    different contexts never order against each other).  Every rank passes its contexts in the same order."""
    from ._abi import load_library
    lib = load_library()

    def make_id():
        import ctypes as C
        buf = (C.c_uint8 * KLT_COMM_ID_BYTES)()
        rc = lib.klt_comm_unique_id(buf)
        if rc != 0:
            from ._abi import KltBackendError
            raise KltBackendError("klt_comm_unique_id failed (%d): %s" % (rc, (lib.klt_last_error(None) or b"").decode()))
        return bytes(buf)

    ids = exchange_ids(rank, world, len(ctxs), make_id, path=path)
    for cx, uid in zip(ctxs, ids):
        cx.comm_init(world, rank, uid)


def cleanup_rendezvous(rank, path=None):
    """Rank 0 removes the id file once every rank has joined (call after the first collective)."""
    if rank == 0:
        try:
            os.unlink(path or rendezvous_file())
        except OSError:
            pass


# ------------------------------------------------------------------------------------------ launcher
def spawn_ranks(argv, n, env=None, timeout=None):
    """Start `n` processes of `argv` (one per GPU) with RANK / LOCAL_RANK / WORLD_SIZE / KLT_RDZV_FILE set, forward
    rank 0's stdout, and return the worst exit code.  The CALLER MUST NOT HAVE TOUCHED THE GPU (no HIP call, no
    klt_create): children are fresh processes started with subprocess -- never fork-after-init, never exec."""
    import tempfile
    base = dict(os.environ if env is None else env)
    rdzv_dir = tempfile.mkdtemp(prefix="klt_rdzv_")
    base.update({"WORLD_SIZE": str(n), "KLT_RDZV_FILE": os.path.join(rdzv_dir, "ids"), "KLT_SPAWNED": "1",
                 "MASTER_ADDR": "127.0.0.1"})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL across processes needs dmabuf IPC on this driver
    procs = []
    out_path = os.path.join(rdzv_dir, "rank0.out")
    t0 = time.monotonic()
    try:
        with open(out_path, "wb") as out0_f:
            for r in range(n):
                e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
                procs.append(subprocess.Popen(list(argv), env=e, stdout=out0_f if r == 0 else subprocess.DEVNULL))
        # a rank that dies would leave the others waiting in a collective: as soon as one fails, stop the rest
        codes = [None] * n
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes) or (timeout is not None and time.monotonic() - t0 > timeout):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.kill()               # exact PIDs we started
                        codes[r] = p.wait()
                if all(c in (None, 0) for c in codes):
                    codes[0] = codes[0] or 124      # timed out
                break
            time.sleep(0.02)
        with open(out_path, "rb") as f:
            out0 = f.read()
    except BaseException:
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise
    finally:
        try:
            for f in os.listdir(rdzv_dir):
                os.unlink(os.path.join(rdzv_dir, f))
            os.rmdir(rdzv_dir)
        except OSError:
            pass
    sys.stdout.buffer.write(out0 or b"")
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    return bad[0] if bad else 0


# ------------------------------------------------------------------------------------- cfg-4 driver
def track_pairs_sharded(track_pair, n_pairs, n_feat, world, rank, gather):
    """cfg-4 driver over a host-level gather: `track_pair(i)` -> [n_feat] FEAT_DTYPE records of pair i (runs on this
    rank's GPU); every rank processes its contiguous shard; `gather(local [pairs_local, n_feat])` returns the
    [n_pairs, n_feat] table in pair order on the destination rank and None elsewhere."""
    mine = shard_range(n_pairs, world, rank)
    local = np.zeros((len(mine), n_feat), FEAT_DTYPE)
    for k, i in enumerate(mine):
        local[k] = track_pair(i)
    return gather(local)


class ShardGather:
    """Device-side gather of a rank's [pairs_local x n_feat] record table to `root` (RCCL, libkltgpu's side stream).

    The table is one feature buffer (`fb_table`) whose rows are `klt_featbuf_view`s the tracker writes into, so a
    shard is gathered with ONE collective.  Shards may differ in size (`shard_range`: 7 or 257 pairs over 8 GPUs):
    the gather carries a count per rank (`klt_gatherv_featbuf_async`); a rank whose shard is empty takes no part."""

    def __init__(self, ctx, fb_table, fb_gathered, n_pairs, n_feat, root=0):
        self.ctx, self.fb_table, self.fb_gathered = ctx, fb_table, fb_gathered
        self.n_pairs, self.n_feat, self.root = n_pairs, n_feat, root
        self.world, self.rank = ctx.comm_info()
        self.counts = [len(shard_range(n_pairs, self.world, r)) * n_feat for r in range(self.world)]
        self.pairs_local = len(shard_range(n_pairs, self.world, self.rank))

    def gather_async(self):
        self.ctx.gatherv_featbuf_async(self.fb_table if self.pairs_local else -1,
                                       self.fb_gathered if self.rank == self.root else -1, self.counts, self.root)

    def result(self):
        """[n_pairs, n_feat] records in pair order on the root (synchronises), None elsewhere."""
        self.ctx.comm_wait()
        if self.rank != self.root:
            return None
        return self.ctx.featbuf_download(self.fb_gathered, self.n_pairs * self.n_feat).reshape(self.n_pairs, self.n_feat)


__all__ = ["shard_range", "world_from_env", "rendezvous_file", "exchange_ids", "init_communicators", "cleanup_rendezvous",
           "spawn_ranks", "track_pairs_sharded", "ShardGather", "KLT_COMM_ID_BYTES"]
