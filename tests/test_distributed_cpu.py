"""world_size-2 gloo tests (CPU) of the N > 1 path: contiguous frame shards + all-gather of per-frame sizes.

The codec itself needs an MI355X, so the per-rank compressor is replaced here by the CPU oracle (allowed in
tests): what is exercised is the sharding arithmetic and the collective, i.e. everything bench.py --gpus N adds."""
import hashlib
import os
import socket
import sys

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


# ---- torch-free: the file communicator (control channel of bench.py --gpus N) and the rank launcher -------------------------

def _file_worker(rank, world, directory, n_frames, q, out_dir):
    from oracle import pyoracle
    from tinyimgcodec_amd.distributed import FileComm, compress_sharded

    comm = FileComm(rank, world, directory=directory, name="t", timeout_s=60)
    got = comm.all_gather_u64(np.arange(5, dtype=np.uint64) + np.uint64(100 * rank))
    assert got.shape == (world, 5) and all(np.array_equal(got[r], np.arange(5, dtype=np.uint64) + np.uint64(100 * r)) for r in range(world))
    assert list(comm.allreduce_max([float(rank), -float(rank), 7.0])) == [world - 1.0, 0.0, 7.0]
    for _ in range(20):  # many collectives: files of finished ones are removed as the ranks go
        comm.barrier()
    assert len([f for f in os.listdir(directory) if f.endswith("_%d" % rank) and f.startswith("t_")]) <= 2

    def get_frame(i):
        return rand_frame(1234 + i, 40 + 8 * (i % 3), 64)

    lo, hi, streams, sizes, offsets = compress_sharded(get_frame, n_frames, q, comm=comm,
                                                       compress_batch_fn=lambda fr, quality: [pyoracle.compress(f, quality) for f in fr])
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, sizes=sizes, offsets=offsets,
             digest=np.frombuffer(hashlib.sha256(b"".join(streams)).digest(), dtype=np.uint8))
    comm.close()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 8)])
def test_file_communicator_shard_and_size_gather(tmp_path, oracle, world, n_frames):
    """The same flow as the gloo test above on distributed.FileComm (tic_rdv_publish / tic_rdv_wait of the C-ABI): no torch,
    no sockets, `world` plain processes."""
    import multiprocessing as mp

    q = 50
    d = tmp_path / "rdv"
    d.mkdir(mode=0o700)
    out = tmp_path / "out"
    out.mkdir()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_file_worker, args=(r, world, str(d), n_frames, q, str(out))) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    want = [oracle.compress(rand_frame(1234 + i, 40 + 8 * (i % 3), 64), q) for i in range(n_frames)]
    want_sizes = np.array([len(s) for s in want], dtype=np.int64)
    covered = []
    for r in range(world):
        x = np.load(os.path.join(str(out), "r%d.npz" % r))
        assert np.array_equal(x["sizes"], want_sizes) and np.array_equal(x["offsets"], np.concatenate([[0], np.cumsum(want_sizes)]))
        lo, hi = int(x["lo"]), int(x["hi"])
        covered.extend(range(lo, hi))
        assert bytes(x["digest"]) == hashlib.sha256(b"".join(want[lo:hi])).digest()
    assert covered == list(range(n_frames))


def test_file_communicator_times_out_on_a_missing_rank_and_refuses_foreign_directories(tmp_path):
    from tinyimgcodec_amd import _native as N
    from tinyimgcodec_amd.distributed import FileComm

    d = tmp_path / "rdv"
    d.mkdir(mode=0o700)
    c = FileComm(0, 2, directory=str(d), name="lonely", timeout_s=0.3)
    with pytest.raises(N.NativeError, match="did not hear from rank 1"):
        c.barrier()
    open_dir = tmp_path / "open"
    open_dir.mkdir()
    os.chmod(str(open_dir), 0o777)
    with pytest.raises(RuntimeError, match="not a private directory"):
        FileComm(0, 1, directory=str(open_dir))


_CHILD = r"""
import os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.path.isdir(os.environ["TIC_RDV_DIR"])
mode = sys.argv[1]
print("rank %d of %d says hello" % (rank, world), flush=True)
open(os.path.join(sys.argv[2], "pid%d" % rank), "w").write(str(os.getpid()))
if mode == "ok":
    sys.exit(0)
if mode == "fail1" and rank == 1:
    t_end = time.time() + 20  # (every rank has come up - on a busy host an interpreter can take longer to start than this one to fail)
    while time.time() < t_end and not all(os.path.exists(os.path.join(sys.argv[2], "pid%d" % r)) and os.path.getsize(os.path.join(sys.argv[2], "pid%d" % r)) for r in range(world)):
        time.sleep(0.02)
    time.sleep(0.1)
    sys.exit(3)
time.sleep(600)   # "hang": everybody in mode hang, the survivors in mode fail1 / killed
"""


def _run_launcher(tmp_path, mode, world=3, timeout_s=30.0, killer=None):
    import io
    import threading
    import time

    from tinyimgcodec_amd.launch import run_ranks

    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    out, err = io.StringIO(), io.StringIO()
    for f in tmp_path.glob("pid*"):  # (an earlier launch's files)
        f.unlink()
    if killer is not None:
        threading.Thread(target=killer, daemon=True).start()
    t0 = time.monotonic()
    rc = run_ranks([sys.executable, str(script), mode, str(tmp_path)], world, timeout_s=timeout_s, stdout=out, stderr=err, grace_s=2.0)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(world)]
    time.sleep(0.1)
    for pid in pids:  # nobody the launcher started is left behind
        assert not os.path.exists("/proc/%d" % pid) or open("/proc/%d/stat" % pid).read().split(")")[1].split()[0] == "Z", pid
    return rc, out.getvalue(), err.getvalue(), time.monotonic() - t0


def test_launcher_relays_rank0_and_reports_failures(tmp_path):
    """tinyimgcodec_amd/launch.py, what `python bench.py --gpus N` runs its ranks with: rank 0's stdout is the job's, a rank
    that fails (or is killed from outside) takes the job down with a non-zero code well inside the watchdog, a hang ends at the
    watchdog with 124, and no rank process survives the launcher."""
    import signal
    import time

    rc, out, err, dt = _run_launcher(tmp_path, "ok")
    assert rc == 0 and out == "rank 0 of 3 says hello\n" and "[rank 1] rank 1 of 3 says hello" in err and "[rank 2]" in err
    rc, out, err, dt = _run_launcher(tmp_path, "fail1")
    assert rc == 3 and dt < 10 and "rank 1 exited with 3" in err

    def killer():  # one rank dies mid-run (OOM killer, a GPU fault): SIGKILL to its exact pid
        p = tmp_path / "pid2"
        while not p.exists() or not p.read_text():
            time.sleep(0.05)
        time.sleep(0.2)
        os.kill(int(p.read_text()), signal.SIGKILL)

    rc, out, err, dt = _run_launcher(tmp_path, "hang", killer=killer)
    assert rc == 128 + signal.SIGKILL and dt < 10 and "rank 2 exited with -9" in err
    rc, out, err, dt = _run_launcher(tmp_path, "hang", world=2, timeout_s=3.0)
    assert rc == 124 and dt < 10 and "still running" in err
    assert not [d for d in os.listdir("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") if d.startswith("tic_rdv_") and
                os.stat(os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", d)).st_uid == os.geteuid() and
                not d.startswith("tic_rdv_%d_" % os.geteuid())]  # every launch removed its rendezvous directory


def test_bench_gpus_n_launches_itself_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` as the driver types it: no launcher around it, no torch.  Here (no GPU) every rank fails at
    tic_create - the product has no CPU fallback - and the parent must exit non-zero, promptly, with the reason on stderr.
    (On the GPU box the same command runs the two-rank rehearsal: tests/test_gpu_parity.py.)"""
    import subprocess

    from tinyimgcodec_amd import _native as N

    if N.load().tic_device_count() > 0:
        pytest.skip("a GPU is present: the launch is exercised by the gpu-marked rehearsal test")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-timeout", "120"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "tic_create" in r.stderr and "stopping the other ranks" in r.stderr
