"""Multi-GPU sharding of independent frame pairs (BASELINE cfg-4) -- one process per GPU.

The KLT path shards by frame pair: every pair is independent, so ranks never exchange pixels or
pyramids.  The only communication is the gather of the 16-byte feature records at the end of a
step (n_feat * 16 B per pair; 80 KB for 5000 features), done with one collective over
RCCL/xGMI (`torch.distributed` backend "nccl") -- or gloo on CPU in the tests.  torch is only
plumbing here (process group + collective); it is imported by the caller and passed in, so the
single-GPU product path never imports it.
"""
import numpy as np

from .backend import FEAT_DTYPE


def shard_range(n_items, world, rank):
    """Contiguous block of `n_items` owned by `rank` (blocks differ by at most one item)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


class _DeviceArray:
    """Zero-copy view of a device allocation for torch.as_tensor (CUDA array interface v2)."""
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class FeatureGather:
    """All-gather of device-resident feature buffers, overlapped with the next step's kernels.

    The collective runs on a side stream that waits (event) for the tracker launch that produced the buffer; the
    tracker's own stream never waits for RCCL, except that a buffer is not overwritten before the gather that reads
    it has finished (`wait_free`).  No host synchronisation anywhere."""

    def __init__(self, ctx, fbs, n, world, torch, dist):
        """fbs: feature buffers of n records each (n may be frames * features of a device-side table)."""
        self.torch, self.dist, self.n, self.world = torch, dist, n, world
        dev = torch.device("cuda", ctx.device)
        self.stream = torch.cuda.ExternalStream(ctx.track_stream_handle(), device=dev)   # the stream the tracker runs on
        self.side = torch.cuda.Stream(device=dev)
        self.views, self.outs, self.ready, self.done = {}, {}, {}, {}
        for fb in fbs:
            ptr = ctx.featbuf_devptr(fb)
            if not ptr:
                raise ValueError("feature buffer %d is not allocated" % fb)
            self.views[fb] = torch.as_tensor(_DeviceArray(ptr, (n, 4), "<i4"), device=dev)
            self.outs[fb] = torch.empty((world, n, 4), dtype=torch.int32, device=dev)
            self.ready[fb] = torch.cuda.Event()
            self.done[fb] = None
        self.last = None

    def wait_free(self, fb):
        """Call before enqueueing work that overwrites buffer `fb`."""
        if self.done[fb] is not None:
            self.stream.wait_event(self.done[fb])

    def all_gather(self, fb):
        """Call right after enqueueing the tracker launch that fills buffer `fb`."""
        self.ready[fb].record(self.stream)
        self.side.wait_event(self.ready[fb])
        with self.torch.cuda.stream(self.side):
            self.dist.all_gather_into_tensor(self.outs[fb], self.views[fb])
            ev = self.torch.cuda.Event()
            ev.record(self.side)
        self.done[fb] = ev
        self.last = fb

    def result(self):
        """[world, n] structured records of the most recent gather, on the host (synchronises)."""
        self.side.synchronize()
        return self.outs[self.last].cpu().numpy().view(FEAT_DTYPE).reshape(self.world, self.n)


def gather_records_host(local, world, torch, dist, dst=0):
    """Gather per-rank record arrays [pairs_local, n] (FEAT_DTYPE) to `dst` through host tensors
    (gloo, or nccl with staging).  Ranks may own different numbers of pairs.  Returns
    [pairs_total, n] on `dst` (rank order = pair order for contiguous shards), None elsewhere."""
    local = np.ascontiguousarray(local, FEAT_DTYPE)
    counts = [None] * world
    dist.all_gather_object(counts, int(local.shape[0]))
    n = local.shape[1]
    width = max(counts)
    pad = np.zeros((width, n), FEAT_DTYPE)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad.view(np.int32).reshape(width, n, 4))
    outs = [torch.empty_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t, outs, dst=dst)
    if outs is None:
        return None
    parts = [o.numpy().reshape(width, n * 4).view(FEAT_DTYPE).reshape(width, n)[:c] for o, c in zip(outs, counts)]
    return np.concatenate(parts, axis=0)


def track_pairs_sharded(track_pair, n_pairs, n_feat, world, rank, torch, dist, dst=0):
    """cfg-4 driver: `track_pair(i)` -> [n_feat] FEAT_DTYPE records of pair i (runs on this rank's GPU).
    Every rank processes its contiguous shard; the records are gathered to `dst` in pair order."""
    mine = shard_range(n_pairs, world, rank)
    local = np.zeros((len(mine), n_feat), FEAT_DTYPE)
    for k, i in enumerate(mine):
        local[k] = track_pair(i)
    return gather_records_host(local, world, torch, dist, dst=dst)
