"""world_size-2 gloo tests (CPU) of the N > 1 path: contiguous frame shards + all-gather of per-frame sizes.

The codec itself needs an MI355X, so the per-rank compressor is replaced here by the CPU oracle (allowed in
tests): what is exercised is the sharding arithmetic and the collective, i.e. everything bench.py --gpus N adds."""
import hashlib
import os
import socket

import numpy as np
import pytest

from conftest import rand_frame

from tinyimgcodec_amd.distributed import shard_range


def test_shard_range_partitions_every_batch():
    for n in (0, 1, 2, 7, 8, 9, 255, 256, 2048):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got.extend(range(lo, hi))
            assert got == list(range(n)), (n, world)
    assert shard_range(2048, 3, 8) == (768, 1024)  # BASELINE config 4: 256 frames per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from tinyimgcodec_amd.distributed import TorchComm, compress_sharded

    called = []

    def get_frame(i):
        called.append(i)
        return rand_frame(1234 + i, 40 + 8 * (i % 3), 64)

    def cpu_batch(frames, quality):
        return [pyoracle.compress(f, quality) for f in frames]

    comm = TorchComm()
    assert (comm.rank, comm.world) == (rank, world)
    lo, hi, streams, sizes, offsets = compress_sharded(get_frame, n_frames, q, comm=comm, compress_batch_fn=cpu_batch)
    assert float(comm.allreduce_max([float(rank)])[0]) == world - 1  # bench.py's max-over-ranks
    assert called == list(range(lo, hi))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, sizes=sizes, offsets=offsets,
             digest=np.frombuffer(hashlib.sha256(b"".join(streams)).digest(), dtype=np.uint8))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 8, 1])
def test_two_rank_gloo_shard_and_size_gather(tmp_path, oracle, n_frames):
    import torch.multiprocessing as mp

    world, q = 2, 50
    mp.spawn(_worker, args=(world, _free_port(), n_frames, q, str(tmp_path)), nprocs=world, join=True)
    want = [oracle.compress(rand_frame(1234 + i, 40 + 8 * (i % 3), 64), q) for i in range(n_frames)]
    want_sizes = np.array([len(s) for s in want], dtype=np.int64)
    covered = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(d["sizes"], want_sizes)  # every rank knows every frame's size
        assert np.array_equal(d["offsets"], np.concatenate([[0], np.cumsum(want_sizes)]))
        lo, hi = int(d["lo"]), int(d["hi"])
        covered.extend(range(lo, hi))
        assert bytes(d["digest"]) == hashlib.sha256(b"".join(want[lo:hi])).digest()
    assert covered == list(range(n_frames))
