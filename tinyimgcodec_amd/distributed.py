"""Multi-GPU sharding of a batch of independent frames (BASELINE configs 3/4).

The path shards embarrassingly: frame i of a batch of B goes to rank i // ceil(B/G) (contiguous shards), every rank
runs the single-GPU pipeline on its own shard, and there is NO data-path collective.  The only exchange is the one
the north star names: an all-gather of per-frame compressed sizes so that every rank knows every output offset.

Three communicators with the same small interface (rank, world, all_gather_u64, allreduce_max, barrier, close):

  RcclComm   the product: RCCL over xGMI through the C-ABI (tic_comm_create_ex / tic_gather_sizes, include/
             tinyimgcodec_hip.h) - one process per GPU, no torch anywhere.  Latency-bound: 8 bytes per frame.
  FileComm   the torch-free CONTROL channel of a one-process-per-GPU job on one node: every collective is one small file per
             rank in a directory private to the launch (tic_rdv_publish / tic_rdv_wait of the same C-ABI; no GPU, no sockets).
             The ranks use it to agree that RCCL came up EVERYWHERE before any of them enters an RCCL collective (a rank whose
             communicator failed would otherwise leave the others inside one without a timeout), and it carries the whole
             exchange where RCCL cannot run (several ranks rehearsing on one GPU).  Never on the data path of a real run.
  TorchComm  torch.distributed plumbing (gloo on CPU) for callers that already live inside a torch.distributed job, and for
             the world_size-2 gloo tests.  Nothing in the product or in bench.py imports torch.

Launching: `tinyimgcodec_amd.launch.run_ranks` (what `python bench.py --gpus N` uses) starts one fresh process per GPU with
RANK / LOCAL_RANK / WORLD_SIZE / TIC_RDV_DIR in the environment; under torch.distributed.run the same variables come from
torchrun and the rendezvous directory is derived from the launcher's identity (rendezvous_dir()).
"""
import ctypes as C
import os
import stat
import tempfile
import time

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


def launcher_start_realtime_ns():
    """CLOCK_REALTIME at which the launcher (the parent process) started, 0 if /proc cannot tell: the `not_before` of every
    rendezvous file of the launch - a rank started long after rank 0 published (a staggered launcher, a restarted worker)
    still believes rank 0's file, while anything an EARLIER launch left under the same name is ignored."""
    ticks = _launcher_start_ticks()
    try:
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        age = up - ticks / float(os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return 0
    if ticks == 0 or age < 0:
        return 0
    return max(2, int((time.time() - age - 2.0) * 1e9))  # (two seconds for the clock-tick granularity)


def _private_dir(path):
    """Creates (mode 0700) or accepts `path` as a directory owned by this user and closed to everybody else."""
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user" % path)
    return path


def rendezvous_dir():
    """Directory of this launch's rendezvous files, the same on every rank: TIC_RDV_DIR (set by launch.run_ranks, which creates it
    with mkdtemp and removes it afterwards), otherwise <tmp>/tic_rdv_<uid>_<MASTER_PORT>_<launcher pid>_<launcher start time>
    - the ranks of one torchrun are children of the same launcher - created 0700 by whichever rank comes first and checked for
    ownership and mode by all of them."""
    d = os.environ.get("TIC_RDV_DIR")
    if d:
        return _private_dir(d)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    return _private_dir(os.path.join(base, "tic_rdv_%d_%s_%d_%d" % (os.geteuid(), os.environ.get("MASTER_PORT", "0"), os.getppid(),
                                                                   _launcher_start_ticks())))


def default_rendezvous_path():
    """A file name unique to one launch of a one-process-per-GPU job and to one communicator of that job: the ranks share the
    launch's private directory and create their communicators in the same order."""
    global _rdv_seq
    _rdv_seq += 1
    return os.path.join(rendezvous_dir(), "rccl_id_%d" % _rdv_seq)


def _not_before_ns():
    """Launches through run_ranks use a fresh mkdtemp directory: any file in it belongs to this launch (1 = any age).  Elsewhere
    the launcher's start time bounds what a reader believes."""
    if os.environ.get("TIC_RDV_DIR"):
        return 1
    return launcher_start_realtime_ns()


class FileComm:
    """Collectives over small files in the launch's private directory (see the module docstring): collective number k of
    communicator `name` is the file <dir>/<name>_<k>_<rank> of every rank - publish mine, read everybody's.  Once collective k
    has completed on a rank, every rank has published k, i.e. has finished reading k - 1: the rank removes its file of k - 1.
    A rank that never arrives makes the others raise after `timeout_s`.  (The empty file of the closing barrier stays behind: a
    slower rank may still be reading it.  run_ranks removes the whole directory; elsewhere it is one empty file per rank.)"""

    def __init__(self, rank=None, world=None, directory=None, name="ctl", timeout_s=180.0):
        self._L = N.load()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        self.dir = _private_dir(directory) if directory else rendezvous_dir()
        self.name, self.timeout_ms = name, int(timeout_s * 1000)
        self._not_before = 1 if (directory or os.environ.get("TIC_RDV_DIR")) else launcher_start_realtime_ns()
        self._k = 0
        self._mine = []

    def _path(self, k, r):
        return os.path.join(self.dir, "%s_%d_%d" % (self.name, k, r)).encode()

    def _exchange(self, payload):
        """payload: bytes of the same length on every rank -> list of every rank's payload."""
        self._k += 1
        k, n = self._k, len(payload)
        buf = C.create_string_buffer(payload, n) if n else None
        rc = self._L.tic_rdv_publish(self._path(k, self.rank), buf, n)
        if rc != N.TIC_OK:
            raise N.NativeError(rc, self._L.tic_comm_last_error(None).decode())
        self._mine.append(k)
        out = []
        for r in range(self.world):
            if r == self.rank:
                out.append(bytes(payload))
                continue
            got = C.create_string_buffer(max(n, 1))
            rc = self._L.tic_rdv_wait(self._path(k, r), got if n else None, n, self.timeout_ms, self._not_before)
            if rc != N.TIC_OK:
                raise N.NativeError(rc, "rank %d: collective %d of '%s' did not hear from rank %d: %s"
                                    % (self.rank, k, self.name, r, self._L.tic_comm_last_error(None).decode()))
            out.append(got.raw[:n])
        while self._mine and self._mine[0] < k:  # everybody has published k, so everybody is done reading k - 1
            self._unlink(self._mine.pop(0))
        return out

    def _unlink(self, k):
        try:
            os.unlink(self._path(k, self.rank))
        except OSError:
            pass

    def all_gather_u64(self, mine):
        mine = np.ascontiguousarray(mine, dtype=np.uint64)
        parts = self._exchange(mine.tobytes())
        return np.stack([np.frombuffer(p, dtype=np.uint64) for p in parts]).reshape(self.world, mine.size)

    def allreduce_max(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        parts = self._exchange(v.tobytes())
        return np.max(np.stack([np.frombuffer(p, dtype=np.float64) for p in parts]), axis=0)

    def barrier(self):
        self._exchange(b"")

    def close(self):
        """A closing barrier, after which only its own (empty) file of every rank is left."""
        if self._mine:
            try:
                self._exchange(b"")   # after this, nobody reads anything older than the final (empty) file
            except Exception:  # noqa: BLE001
                pass
            while self._mine:
                k = self._mine.pop(0)
                if k < self._k:
                    self._unlink(k)


class RcclComm:
    """RCCL communicator of one rank (one process per GPU), created on the codec context's device and stream."""

    def __init__(self, ctx, rank=None, world=None, rendezvous_path=None, timeout_s=120.0):
        self._L = N.load()
        self.ctx = ctx
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        path = rendezvous_path or default_rendezvous_path()
        h = C.c_void_p()
        rc = self._L.tic_comm_create_ex(ctx.handle, self.rank, self.world, path.encode(), _not_before_ns(), int(timeout_s * 1000), C.byref(h))
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

    def version(self):
        """NCCL_VERSION_CODE of the RCCL library in use (0 for a single rank: no library is loaded)."""
        return int(self._L.tic_comm_rccl_version(self._h))

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
