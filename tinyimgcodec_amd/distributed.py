"""Multi-GPU sharding of a batch of independent frames (BASELINE configs 3/4).

The path shards embarrassingly: frame i of a batch of B goes to rank i // ceil(B/G) (contiguous shards), every rank
runs the single-GPU pipeline on its own shard, and there is NO data-path collective.  The only exchange is the one
the north star names: an all-gather of per-frame compressed sizes so that every rank knows every output offset.
With the "nccl" backend that all-gather is RCCL over xGMI (latency-bound: 8 bytes per frame); the same code runs on
"gloo" for CPU tests.  torch.distributed is plumbing only - the codec itself never touches torch.
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; shards differ in size by at most one chunk at the tail."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_frames // world) if n_frames else 0
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return lo, hi


def gather_sizes(local_sizes, n_frames, group=None, device=None):
    """All-gather the per-frame compressed sizes of every rank.

    Returns (sizes int64[n_frames], offsets int64[n_frames + 1]) identical on all ranks; offsets are the byte
    positions of each frame in the concatenation of all streams in frame order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_frames // world) if n_frames else 0
    lo, hi = shard_range(n_frames, rank, world)
    if len(local_sizes) != hi - lo:
        raise ValueError("rank %d holds %d sizes for a shard of %d frames" % (rank, len(local_sizes), hi - lo))
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.full((max(per, 1),), -1, dtype=torch.int64, device=device)
    if hi > lo:
        mine[: hi - lo] = torch.as_tensor(np.asarray(local_sizes, dtype=np.int64), device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allsz = torch.cat(parts).cpu().numpy()
    sizes = np.concatenate([allsz[r * max(per, 1) : r * max(per, 1) + (shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0])] for r in range(world)]) if n_frames else np.zeros(0, np.int64)
    if (sizes < 0).any():
        raise RuntimeError("size gather returned an unfilled slot")
    offsets = np.zeros(n_frames + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    return sizes.astype(np.int64), offsets


def compress_sharded(get_frame, n_frames, quality=50, compress_batch_fn=None, group=None, threads=0):
    """Compress this rank's shard of a batch of n_frames frames and gather all sizes.

    get_frame(i) -> 2-D uint8 array of frame i (only called for this rank's frames).
    compress_batch_fn(frames, quality) -> list[bytes]; defaults to the MI355X pipeline (tinyimgcodec_amd.compress_batch).
    Returns (lo, hi, streams_of_this_rank, sizes_of_all_frames, offsets_of_all_frames)."""
    import torch.distributed as dist

    if compress_batch_fn is None:
        from .codec import compress_batch

        def compress_batch_fn(frames, q):
            return compress_batch(frames, q, threads=threads)

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_frames, rank, world)
    frames = [get_frame(i) for i in range(lo, hi)]
    streams = compress_batch_fn(frames, quality) if frames else []
    sizes, offsets = gather_sizes([len(s) for s in streams], n_frames, group=group)
    return lo, hi, streams, sizes, offsets
