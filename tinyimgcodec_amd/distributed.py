"""Multi-GPU sharding of a batch of independent frames (BASELINE configs 3/4).

The path shards embarrassingly: frame i of a batch of B goes to rank i // ceil(B/G) (contiguous shards), every rank
runs the single-GPU pipeline on its own shard, and there is NO data-path collective.  The only exchange is the one
the north star names: an all-gather of per-frame compressed sizes so that every rank knows every output offset.

Two communicators with the same small interface (rank, world, all_gather_u64, allreduce_max, barrier, close):

  RcclComm   the product: RCCL over xGMI through the C-ABI (tic_comm_create / tic_gather_sizes, include/
             tinyimgcodec_hip.h) - one process per GPU, no torch anywhere.  Latency-bound: 8 bytes per frame.
  TorchComm  torch.distributed plumbing (gloo on CPU) for the world_size-2 tests that run where no GPU exists, and for
             callers that already live inside a torch.distributed job.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N


def shard_range(n_frames, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; shards differ in size by at most one chunk at the tail."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_frames // world) if n_frames else 0
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return lo, hi


_rdv_seq = 0


def _launcher_start_ticks():
    """Start time (clock ticks since boot) of the launcher process: with its pid it names ONE launcher instance, pid reuse or not."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


def default_rendezvous_path():
    """A file name unique to one launch of a one-process-per-GPU job and to one communicator of that job: the ranks are
    children of the same launcher (same pid, same start time) and create their communicators in the same order."""
    global _rdv_seq
    _rdv_seq += 1
    return "/tmp/tic_rdv_%s_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), _launcher_start_ticks(), _rdv_seq)


class RcclComm:
    """RCCL communicator of one rank (one process per GPU), created on the codec context's device and stream."""

    def __init__(self, ctx, rank=None, world=None, rendezvous_path=None):
        self._L = N.load()
        self.ctx = ctx
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        path = rendezvous_path or default_rendezvous_path()
        h = C.c_void_p()
        rc = self._L.tic_comm_create(ctx.handle, self.rank, self.world, path.encode(), C.byref(h))
        if rc != N.TIC_OK:
            raise N.NativeError(rc, self._L.tic_comm_last_error(None).decode())
        self._h = h

    def _check(self, rc):
        if rc != N.TIC_OK:
            raise N.NativeError(rc, self._L.tic_comm_last_error(self._h).decode())

    def all_gather_u64(self, mine):
        """mine: uint64[n] (same n on every rank) -> uint64[world, n]."""
        mine = np.ascontiguousarray(mine, dtype=np.uint64)
        out = np.empty((self.world, mine.size), dtype=np.uint64)
        self._check(self._L.tic_gather_sizes(self._h, mine.ctypes.data, int(mine.size), out.ctypes.data))
        return out

    def allreduce_max(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64).copy()
        self._check(self._L.tic_comm_allreduce_max(self._h, v.ctypes.data, int(v.size)))
        return v

    def barrier(self):
        self.allreduce_max([0.0])

    def close(self):
        if self._h:
            self._L.tic_comm_destroy(self._h)
            self._h = None


class TorchComm:
    """The same interface on torch.distributed (gloo on CPU; with nccl the tensors live on cuda:LOCAL_RANK)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device = device

    def all_gather_u64(self, mine):
        torch, dist = self._torch, self._dist
        mine = np.ascontiguousarray(mine, dtype=np.uint64)
        t = torch.as_tensor(mine.view(np.int64), device=self.device)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return torch.stack(parts).cpu().numpy().view(np.uint64)

    def allreduce_max(self, vals):
        torch, dist = self._torch, self._dist
        t = torch.as_tensor(np.ascontiguousarray(vals, dtype=np.float64), device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def barrier(self):
        self._dist.barrier(self.group)

    def close(self):
        pass


_UNFILLED = np.uint64(0xFFFFFFFFFFFFFFFF)


def gather_sizes(local_sizes, n_frames, comm):
    """All-gather the per-frame compressed sizes of every rank.

    Returns (sizes int64[n_frames], offsets int64[n_frames + 1]) identical on all ranks; offsets are the byte
    positions of each frame in the concatenation of all streams in frame order."""
    world, rank = comm.world, comm.rank
    per = -(-n_frames // world) if n_frames else 0
    lo, hi = shard_range(n_frames, rank, world)
    if len(local_sizes) != hi - lo:
        raise ValueError("rank %d holds %d sizes for a shard of %d frames" % (rank, len(local_sizes), hi - lo))
    mine = np.full(max(per, 1), _UNFILLED, dtype=np.uint64)
    if hi > lo:
        mine[: hi - lo] = np.asarray(local_sizes, dtype=np.uint64)
    allsz = comm.all_gather_u64(mine)
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(n_frames, r, world)
        parts.append(allsz[r, : rhi - rlo])
    sizes = np.concatenate(parts) if n_frames else np.zeros(0, np.uint64)
    if (sizes == _UNFILLED).any():
        raise RuntimeError("size gather returned an unfilled slot")
    sizes = sizes.astype(np.int64)
    offsets = np.zeros(n_frames + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    return sizes, offsets


def compress_sharded(get_frame, n_frames, quality=50, comm=None, compress_batch_fn=None, threads=0):
    """Compress this rank's shard of a batch of n_frames frames and gather all sizes.

    get_frame(i) -> 2-D uint8 array of frame i (only called for this rank's frames).
    comm: RcclComm (one process per GPU) or TorchComm.
    compress_batch_fn(frames, quality) -> list[bytes]; defaults to the MI355X pipeline (tinyimgcodec_amd.compress_batch).
    Returns (lo, hi, streams_of_this_rank, sizes_of_all_frames, offsets_of_all_frames)."""
    if comm is None:
        raise ValueError("compress_sharded needs a communicator (RcclComm or TorchComm)")
    if compress_batch_fn is None:
        from .codec import compress_batch

        def compress_batch_fn(frames, q):
            return compress_batch(frames, q, threads=threads)

    lo, hi = shard_range(n_frames, comm.rank, comm.world)
    frames = [get_frame(i) for i in range(lo, hi)]
    streams = compress_batch_fn(frames, quality) if frames else []
    sizes, offsets = gather_sizes([len(s) for s in streams], n_frames, comm)
    return lo, hi, streams, sizes, offsets
