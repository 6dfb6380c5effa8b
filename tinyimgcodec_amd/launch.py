"""One fresh process per GPU, torch-free: what `python bench.py --gpus N` (N > 1) does when it was not started by a launcher.

run_ranks() must be called BEFORE anything in the calling process has touched HIP (it only spawns children; it never execs over
itself): every rank is a fresh interpreter with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / TIC_RDV_DIR set, the same
variables torch.distributed.run would export (MASTER_ADDR / MASTER_PORT are set too, for code that wants them; nothing here
opens a socket).  TIC_RDV_DIR is a mkdtemp directory (0700) that holds the launch's rendezvous files - the RCCL unique id and
the file communicator's collectives (distributed.py) - and is removed when the launch ends.

Failure handling: the launcher waits for all ranks; as soon as one exits non-zero, or when `timeout_s` expires, the others are
terminated (SIGTERM to the exact pids, SIGKILL after a grace period) and the launcher reports a non-zero code.  Rank 0's stdout
is the job's stdout (bench.py prints its one JSON line there); the other ranks' stdout goes to stderr, prefixed.
"""
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time


def _relay(stream, sink, prefix):
    for line in iter(stream.readline, b""):
        sink.write(prefix + line.decode(errors="replace"))
        sink.flush()
    stream.close()


def run_ranks(cmd, world, timeout_s=1500.0, extra_env=None, grace_s=5.0, stdout=None, stderr=None):
    """Starts `cmd` (argv list) `world` times, one rank each; returns 0 if every rank exited 0, else the first failing rank's
    code (124 after a timeout, as timeout(1) reports it)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    rdv = tempfile.mkdtemp(prefix="tic_rdv_", dir=base)
    procs, threads = [], []
    rc = 0
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                        "TIC_RDV_DIR": rdv, "MASTER_ADDR": "127.0.0.1"})
            env.setdefault("MASTER_PORT", "29500")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if extra_env:
                env.update(extra_env)
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None if stderr is sys.stderr else subprocess.PIPE)
            procs.append(p)
            t = threading.Thread(target=_relay, args=(p.stdout, stdout if r == 0 else stderr, "" if r == 0 else "[rank %d] " % r), daemon=True)
            t.start()
            threads.append(t)
            if p.stderr is not None:
                t = threading.Thread(target=_relay, args=(p.stderr, stderr, ""), daemon=True)
                t.start()
                threads.append(t)
        deadline = time.monotonic() + timeout_s
        live = set(range(world))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 128 - code  # (killed by signal s: 128 + s, as a shell reports it)
                        stderr.write("launch: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                        break
            if rc == 0 and live and time.monotonic() > deadline:
                rc = 124
                stderr.write("launch: ranks %s still running after %.0f s; stopping them\n" % (sorted(live), timeout_s))
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        _stop([p for p in procs if p.poll() is None], grace_s)
        for t in threads:
            t.join(timeout=2.0)
        shutil.rmtree(rdv, ignore_errors=True)
    return rc


def _stop(procs, grace_s):
    """Ends exactly the processes this launcher started (by pid, never by pattern)."""
    for p in procs:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    t_end = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()
