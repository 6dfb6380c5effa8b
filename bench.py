#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X tinyimgcodec hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one launch of the transform stage (level shift -> 2-D DCT -> quantise -> zig-zag, i.e. the reference's
encode(), codec.py:26-43) over this rank's input, already resident in HBM.

  N = 1   BASELINE config 2: ONE 4096x4096 random uint8 frame (numpy default_rng(1234)), quality 50, one launch per step.
  N > 1   BASELINE config 4: a batch of 256*N 1920x1080 frames (seed 1234 + i), contiguous shards of 256 frames per
          rank (shard_range), one batched launch of the rank's 256 resident frames per step; afterwards every rank sends its
          shard through the whole pipeline host -> host (tic_compress_batch: pinned staging, H2D || kernels || D2H, entropy
          stage on the device) and the ACTUAL 256 stream sizes of every rank are all-gathered with RCCL (tic_gather_sizes).
          Barrier and max-over-ranks use RCCL too.  No torch anywhere: `python bench.py --gpus N` starts its own N rank
          processes (tinyimgcodec_amd/launch.py; under torch.distributed.run the ranks are torchrun's), and the ranks agree
          over small files in the launch's private directory (distributed.FileComm on tic_rdv_publish / tic_rdv_wait) on
          whether every rank's RCCL communicator came up - that channel carries the whole exchange if one did not.
  --workload config4 runs the N > 1 workload on ONE GPU too (the N = 1 point of a config-4 scaling curve).

SCALING: N = 1 and N > 1 measure DIFFERENT workloads by default (config 2: one 50 MB launch per step, launch gaps included;
config 4: 1.6 GB per launch).  A curve must compare like with like: every N > 1 line names its baseline
(`scaling_baseline`): the N = 1 line's `config4.kernel_only_mpix_s` for `value`, `config4.host_to_host_mpix_s` for
`config.host_to_host_mpix_s` - or the `value` of a `--gpus 1 --workload config4` run.  `value` at N > 1 is kernel-only weak
scaling with no data-path collective (N x by construction, up to clock and power differences between GPUs); the figure that
can fail to scale is host -> host (PCIe, host DRAM, NUMA), reported beside it.

Rank 0 prints ONE JSON line.  `value` = whole-job Mpixel/s = pixels of all ranks x K / max-over-ranks time of the K steps: at
N = 1 the interval between two HIP events on the launch stream around the K launches (the host clock around the same bracket is
reported beside it, `config.wall_ms_per_step`: it adds one submission ramp and one completion wake-up per CALL, 40-70 us, which
at K = 20 would read as 3 us per step), at N > 1 the host clock between the two barriers (K x 280 us).  `roofline.achieved` = algorithmic bytes per launch (3 B/pixel: 1 B read + 2 B written) / average launch duration
measured with HIP events recorded on the library's own stream around the same K launches; `roofline.cold` = the same with
12 distinct frame/coefficient buffer pairs in rotation (604 MB > the 256 MiB Infinity Cache: every launch streams from and
to HBM).  `cpu_baseline` / `cpu_baseline_threads` = the oracle (C restatement of the reference's CPU path) on this host, 1 thread /
row bands on this process's CPU share (at most 16 threads: a GPU box gives one GPU 16 of its cores; `cores` says how many);
`cpu_baseline_numpy_scipy` = oracle/np_encode.py, a numpy/scipy restatement of encode() on the reference's own library stack,
where scipy is importable.  `config4` = the 256-frame shard of this rank, kernel-only and host -> host.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
BYTES_PER_PIXEL = 3.0  # 1 B uint8 read + 2 B int16 written
COLD_PAIRS = 12        # 12 x (16.8 MB frame + 33.6 MB coefficients) = 604 MB in rotation


def rand_frame(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--quality", type=int, default=50)
    ap.add_argument("--variant", choices=["hybrid", "exact"], default="hybrid")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="N=1 only: skip the 256-frame shard measurements")
    ap.add_argument("--shard-frames", type=int, default=256, help="frames per rank of BASELINE config 4")
    ap.add_argument("--workload", choices=["auto", "config2", "config4"], default="auto",
                    help="auto: config 2 at N = 1 (the headline), config 4 at N > 1; config4 at N = 1 = the baseline of a config-4 scaling curve")
    ap.add_argument("--settle-ms", type=float, default=60.0, help="untimed back-to-back launches before the warm-up steps (clock settling)")
    ap.add_argument("--per-launch", choices=["same", "separate", "off"], default="separate",
                    help="per_launch_us {min, median, max} from events on the launches' own dispatch packets: in a pass of its own behind the timed one "
                    "(default: a dispatch that carries events is followed by a 5 us gap, profiles/r06_driver_flags.txt), on the timed launches themselves, or not at all")
    ap.add_argument("--no-preroll", dest="preroll", action="store_false", help="do not put 32 more settling launches into the timed submission")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="--gpus N > 1 without a launcher: watchdog over the N rank processes (s)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process becomes the launcher - N fresh rank processes, one per GPU, rank 0's
        # JSON line relayed, non-zero exit if any rank fails or the watchdog expires.  Nothing here has touched HIP: the
        # package import below loads no shared object (tinyimgcodec_amd/launch.py is standard library only).
        from tinyimgcodec_amd.launch import run_ranks

        sys.exit(run_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout_s=args.launch_timeout))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world

    import tinyimgcodec_amd as T
    from tinyimgcodec_amd import _native as N
    from tinyimgcodec_amd.distributed import FileComm, RcclComm, gather_sizes, shard_range

    L = N.load()
    # The real run: one rank per GPU, RCCL.  TIC_BENCH_BACKEND=file + TIC_BENCH_SHARE_GPU=1 rehearse the multi-rank flow on a
    # one-GPU box: every rank on device 0 (RCCL needs one GPU per rank, so the rehearsal exchanges over the file communicator).
    share_gpu = os.environ.get("TIC_BENCH_SHARE_GPU", "0") == "1"
    ctx = T.Context(0 if share_gpu else local_rank)  # raises loudly if the HIP library / an MI355X is missing
    # One process per GPU shares the node's cores: each rank's pipeline stages pageable frames on (cores / ranks of the node) / 2
    # threads at most, instead of the single-process default of 8 (8 ranks would otherwise start 64 copy threads).
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1:
        ctx.check(L.tic_set_stage_threads(ctx.handle, max(1, min(8, (os.cpu_count() or 8) // max(local_world, 1) // 2))))
    comm = None
    comm_note = "RCCL through the C-ABI (tic_comm_create_ex / tic_gather_sizes / tic_comm_allreduce_max)"
    if world > 1:
        # The file communicator is the control channel: EVERY rank joins it, whatever became of its RCCL communicator, and the
        # ranks agree on one backend.  (Falling back per rank would leave the ranks whose RCCL did come up inside
        # ncclCommInitRank or the first all-reduce without a timeout.)  A rank that hangs inside RCCL never reaches the
        # agreement: the others time out here and exit non-zero, which makes the launcher tear the job down.
        backend = os.environ.get("TIC_BENCH_BACKEND", "rccl")
        rccl, rccl_err = None, ""
        if backend == "rccl" and not share_gpu:
            try:
                rccl = RcclComm(ctx, rank, world)
            except Exception as e:  # noqa: BLE001
                rccl_err = str(e)[:200]
                sys.stderr.write("bench.py rank %d: RCCL communicator failed (%s)\n" % (rank, e))
        ctl = FileComm(rank, world, name="bench_ctl")
        ok_everywhere = float(ctl.allreduce_max([0.0 if rccl is not None else 1.0])[0]) == 0.0
        if ok_everywhere:
            comm = rccl
        else:
            if rccl is not None:
                rccl.close()
            comm = ctl
            if backend != "rccl" or share_gpu:
                comm_note = "file communicator (rehearsal: TIC_BENCH_SHARE_GPU=1 puts every rank on device 0, where RCCL cannot run)"
            else:
                comm_note = "file communicator fallback: the RCCL communicator did not come up on every rank (this rank: %s)" % (rccl_err or "ok")
    q = args.quality
    variant = N.KERNEL_HYBRID if args.variant == "hybrid" else N.KERNEL_EXACT
    multi = world > 1 or args.workload == "config4"
    if args.workload == "config2" and world > 1:
        sys.exit("--workload config2 is a single-GPU workload")
    if args.steps is None:
        args.steps = 200 if multi else 5000
    if args.warmup is None:
        args.warmup = 20 if multi else 2000

    def barrier():
        ctx.check(L.tic_sync(ctx.handle))
        if comm is not None:
            comm.barrier()

    ms = C.c_float(0.0)
    out = None
    if not multi:
        out = bench_config2(args, ctx, L, N, q, variant, barrier, ms)
        if not args.no_config4:
            out["config4"] = shard_measurements(args, ctx, L, N, T, q, variant, 0, args.shard_frames, ms, steps=50)[0]
            out["config4"]["note"] = "the N = 1 baseline of every N > 1 line (scaling_baseline): same 256-frame shard, one GPU"
        if not args.no_config4:
            out["codec_resident"] = codec_resident(args, ctx, L, N, q)
            out["bench_set"] = bench_set(args, ctx, L)
        if not args.no_cpu_baseline:
            img = rand_frame(1234, args.height, args.width)
            out["cpu_baseline"] = cpu_baseline(img, q, args.cpu_seconds, 1)
            out["cpu_baseline_threads"] = cpu_baseline(img, q, args.cpu_seconds, 0)
            out["cpu_baseline_all_cores"] = cpu_baseline(img, q, min(args.cpu_seconds, 4.0), -1)  # every CPU of the node this process may use
            py = cpu_baseline_numpy_scipy(img, q, args.cpu_seconds)
            if py is not None:
                out["cpu_baseline_numpy_scipy"] = py
    else:
        n_total = args.shard_frames * world
        lo, hi = shard_range(n_total, rank, world)
        info, t_wall, kernel_ms, sizes_mine = shard_measurements(args, ctx, L, N, T, q, variant, lo, hi - lo, ms, steps=args.steps,
                                                                 warmup=args.warmup, barrier=barrier, timed=True)
        node_i, ncpu_i = C.c_int(), C.c_int()
        ctx.check(L.tic_numa_info(ctx.handle, C.byref(node_i), C.byref(ncpu_i)))
        pci = L.tic_pci_bus_id(ctx.handle).decode()
        try:
            bus = int(pci.split(":")[1], 16)
        except Exception:  # noqa: BLE001
            bus = 0xFF
        me = [(L.tic_get_stage_threads(ctx.handle) << 32) | ((node_i.value & 0xFFFF) << 16) | ((ncpu_i.value & 0xFF) << 8) | bus]
        if comm is not None:
            red = comm.allreduce_max([t_wall, kernel_ms, info["host_to_host_s"], info["host_to_host_registered_s"]])  # max over ranks
            sizes, offsets = gather_sizes(sizes_mine, n_total, comm)          # RCCL all-gather of the actual stream sizes
            rank_words, _ = gather_sizes(me, world, comm)                     # per-rank placement: threads, NUMA node, PCI bus
        else:  # --gpus 1 --workload config4
            red = [t_wall, kernel_ms, info["host_to_host_s"], info["host_to_host_registered_s"]]
            sizes = np.asarray(sizes_mine, dtype=np.int64)
            offsets = np.concatenate([[0], np.cumsum(sizes)])
            rank_words = np.asarray(me, dtype=np.int64)
        if rank == 0:
            h, w = 1080, 1920
            pixels = float(h) * w * (hi - lo)
            value = pixels * world * args.steps / float(red[0]) / 1e6
            achieved = BYTES_PER_PIXEL * pixels / (float(red[1]) * 1e-3) / 1e9
            out = {
                "metric": "Mpixels/s encode (DCT+quant kernel)",
                "value": round(value, 1),
                "unit": "Mpix/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(float(red[0]) * 1e3 / args.steps, 6),
                "higher_is_better": True,
                "scaling": "weak",
                "value_note": "kernel-only, N x by construction: every rank times its own resident shard, there is no data-path collective; the figure that "
                "can fail to scale is scaling_headline (host -> host: PCIe, host DRAM, NUMA)",
                "scaling_headline": {"metric": "host_to_host_mpix_s", "value": round(pixels * world / float(red[2]) / 1e6, 1), "unit": "Mpix/s",
                                     "registered_input": round(pixels * world / float(red[3]) / 1e6, 1), "per_rank_frames_per_s": round((hi - lo) / float(red[2]), 1),
                                     "baseline": "`config4.host_to_host_mpix_s` of the N = 1 line", "time": "max over ranks of one tic_compress_batch call on the rank's shard (median of 5)"},
                "vs_baseline": None,
                "dtype": "f32+f64" if args.variant == "hybrid" else "f64",
                "data": "synthetic",
                "config": {
                    "workload": "batch of %d random 1920x1080 uint8 frames (seeds 1234+i), quality=%d, contiguous shards of %d frames "
                    "per GPU resident in HBM, one batched launch per step (BASELINE config 4)" % (n_total, q, hi - lo),
                    "kernel": args.variant,
                    "frames_per_step_per_gpu": hi - lo,
                    "sharding": "independent frames, contiguous shards; no data-path collective; all-gather of per-frame stream sizes",
                    "comm": comm_note,
                    "settle_ms": args.settle_ms,
                    "untimed_launches": info["untimed_launches"],
                    "host_to_host_mpix_s": round(pixels * world / float(red[2]) / 1e6, 1),
                    "host_to_host_registered_mpix_s": round(pixels * world / float(red[3]) / 1e6, 1),
                    "host_to_host_note": "whole pipeline per rank (H2D || kernels || D2H, device entropy stage), PCIe included, max over "
                    "ranks; never `value`.  Plain: pageable caller frames staged into pinned memory; registered: the caller's frames "
                    "pinned with tic_host_register and copied from where they lie",
                    "per_rank": {"kernel_only_mpix_s": round(value / world, 1), "host_to_host_mpix_s": round(pixels / float(red[2]) / 1e6, 1),
                                 "host_to_host_registered_mpix_s": round(pixels / float(red[3]) / 1e6, 1)},
                    "numa": info.get("numa"),
                    "ranks": [{"rank": r, "stage_threads": int(v) >> 32, "numa_node": (lambda x: x - 65536 if x >= 32768 else x)((int(v) >> 16) & 0xFFFF),
                               "cpus_of_node": (int(v) >> 8) & 0xFF, "pci_bus": "%02x" % (int(v) & 0xFF)} for r, v in enumerate(rank_words)],
                    "ranks_note": "host threads each rank's pipeline may stage pageable frames on (tic_set_stage_threads: cores / local ranks / 2, at most 8), "
                    "the NUMA node of its GPU and the CPUs of that node in its affinity mask (0-255, saturating), the GPU's PCI bus",
                    "rccl_version": (comm.version() if hasattr(comm, "version") else None),
                    "parity": dict(info["parity"], note="rank 0's shard; every rank checks its own shard and raises on a difference"),
                    "gathered_sizes": {"frames": int(len(sizes)), "total_bytes": int(offsets[-1]), "first": [int(v) for v in sizes[:8]],
                                            "sha256": hashlib.sha256(sizes.astype("<i8").tobytes()).hexdigest()},
                    "device": ctx.arch,
                },
                "scaling_baseline": {
                    "value": "`config4.kernel_only_mpix_s` of the N = 1 line (or `value` of --gpus 1 --workload config4): the same "
                    "256-frame shard on one GPU.  NOT the N = 1 line's `value`, which is config 2 (one 4096x4096 launch per step)",
                    "host_to_host": "`config4.host_to_host_mpix_s` / `config4.host_to_host_registered_mpix_s` of the N = 1 line",
                    "note": "`value` is kernel-only weak scaling without a data-path collective: N x by construction; host -> host is the "
                    "figure that can fail to scale.  No curve had been measured when this was written (no multi-GPU box was available)",
                },
                "roofline": {
                    "bound": "hbm",
                    "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None,
                    "kernel_us": round(float(red[1]) * 1e3, 3),
                    "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * pixels,
                },
            }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    if world > 1 and ctl is not comm:
        ctl.close()
    ctx.close()


def settle(args, ctx, L, fn):
    """Clock settling (untimed, before the W warm-up steps): the chip needs a few ms of back-to-back launches to reach its
    sustained clocks (200 launches after 20 read 15.4 us where 5000 after 2000 read 12.1 us).  Returns the count."""
    n, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < args.settle_ms:
        n += fn()
    return n


def bench_config2(args, ctx, L, N, q, variant, barrier, ms):
    h, w = args.height, args.width
    img = rand_frame(1234, h, w)
    pitch = (w + 255) // 256 * 256
    host = np.zeros((h, pitch), dtype=np.uint8)
    host[:, :w] = img
    nblk = L.tic_num_blocks(h, w)
    # rotating pairs for the cold figure: 12 at 4096^2 (604 MB); frames whose pair alone exceeds the 256 MiB Infinity Cache need
    # only a few - three 16384^2 pairs are 2.4 GB in rotation
    pairs = 1 if args.no_cold else (COLD_PAIRS if 3 * h * w * COLD_PAIRS < (3 << 30) else 3)
    d_imgs, d_outs = (C.c_void_p * pairs)(), (C.c_void_p * pairs)()
    for k in range(pairs):
        a, b = C.c_void_p(), C.c_void_p()
        ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(a)))
        ctx.check(L.tic_dev_alloc(ctx.handle, nblk * 128, C.byref(b)))
        ctx.check(L.tic_memcpy_h2d(ctx.handle, a, host.ctypes.data, host.size))  # (the same pixels in every pair: only the addresses differ)
        d_imgs[k], d_outs[k] = a, b
    d_img, d_out = d_imgs[0], d_outs[0]

    # one untimed launch with the diagnostic counter on: how many blocks leave the fast path on this frame
    fb = C.c_ulonglong(0)
    ctx.check(L.tic_set_stats(ctx.handle, 1))
    ctx.check(L.tic_last_fallback_blocks(ctx.handle, C.byref(fb)))  # resets the counter
    ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, pitch, q, d_out, variant))
    ctx.check(L.tic_last_fallback_blocks(ctx.handle, C.byref(fb)))
    ctx.check(L.tic_set_stats(ctx.handle, 0))

    def burst():
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, variant, 256, C.byref(ms)))
        return 256

    untimed = settle(args, ctx, L, burst)
    barrier()
    t0 = time.perf_counter()
    # W untimed warm-up launches, event, exactly K launches, event - ONE submission on the launch stream (tic_dctq_dev_timed_warm): the
    # first event is stamped when the last warm-up launch retires with the timed launches queued behind it.  (Rounds 1-5 synchronised
    # between warm-up and the first event: recorded on an idle stream it opened the interval with the first launch's submission latency,
    # ~28 us on the driver's box = 1.4 us per step at K = 20, profiles/r06_driver_flags.txt.)  Returns after the stream has drained.
    pre = 32 if args.preroll else 0  # (256 measured no better and once worse - tools/driver_flags.py `preroll` rows, profiles/r06_driver_flags.txt: the host is 0.3 ms ahead with 32 already)
    if args.per_launch == "same" and args.steps > 32768:
        args.per_launch = "separate"
    per = (C.c_float * (2 * args.steps))() if args.per_launch == "same" else None
    ctx.check(L.tic_dctq_dev_timed_warm(ctx.handle, d_img, h, w, pitch, q, d_out, variant, pre + args.warmup, args.steps, C.byref(ms), per))
    barrier()
    wall_s = time.perf_counter() - t0
    ms_total_timed = ms.value
    kernel_ms = ms_total_timed / args.steps
    parity = coefficient_parity(ctx, L, d_out, h, w, q, nblk)  # the buffer the timed launches just wrote
    per_launch = None
    if args.per_launch != "off":
        kp = args.steps if args.per_launch == "same" else min(args.steps, 1000)
        if per is None:  # a pass of its own behind the timed one, never `value`
            settle(args, ctx, L, burst)  # (the parity check above left the device idle for ~100 ms: clocks settled again, as in front of the timed pass)
            per = (C.c_float * (2 * kp))()
            ctx.check(L.tic_dctq_dev_timed_warm(ctx.handle, d_img, h, w, pitch, q, d_out, variant, pre + args.warmup, kp, C.byref(ms), per))
        dur = [float(per[2 * i]) * 1e3 for i in range(kp)]
        end = [float(per[2 * i + 1]) * 1e3 for i in range(kp)]
        gap = sorted(end[i] - end[i - 1] - dur[i] for i in range(1, kp)) or [0.0]
        sd = sorted(dur)
        per_launch = {"min": round(sd[0], 3), "median": round(sd[kp // 2], 3), "max": round(sd[-1], 3), "mean": round(sum(dur) / kp, 3),
                      "gap_median": round(gap[len(gap) // 2], 3), "gap_max": round(gap[-1], 3),
                      "first_start_to_last_end_per_launch": round(end[-1] / kp, 3), "launches": kp,
                      "source": "start / stop events on each launch's own dispatch packet (hipExtLaunchKernelGGL): min / median / max / mean are the KERNEL's "
                      "durations by the packet's time stamps - what rocprofv3's kernel trace reports; a dispatch that carries events is followed by a "
                      "~5 us gap (gap_median), which is why this is " + ("the timed launches themselves" if args.per_launch == "same" else
                      "a pass of its own behind the timed one (same untimed launches in front) and never `value`; ms_per_step - median = what the queue "
                      "adds between two back-to-back launches of the timed pass")}
    cold = None
    if not args.no_cold:
        for _ in range(3):  # settle + warm-up of the rotating variant
            ctx.check(L.tic_dctq_dev_timed_rotating(ctx.handle, d_imgs, d_outs, pairs, h, w, pitch, q, variant, 600 if 3 * h * w < (200 << 20) else 20, C.byref(ms)))
        kcold = max(600, min(args.steps, 3000)) if 3 * h * w < (200 << 20) else max(args.steps, 20)
        ctx.check(L.tic_dctq_dev_timed_rotating(ctx.handle, d_imgs, d_outs, pairs, h, w, pitch, q, variant, kcold, C.byref(ms)))
        cold_ms = ms.value / kcold
        cold_gbs = BYTES_PER_PIXEL * h * w / (cold_ms * 1e-3) / 1e9
        cold = {"achieved": round(cold_gbs, 1), "frac": round(cold_gbs / HBM_PEAK_GBS, 4), "kernel_us": round(cold_ms * 1e3, 3), "launches": kcold,
                "pairs": pairs, "rotating_bytes": int(pairs * (host.size + nblk * 128)),
                "note": "launch i uses frame/coefficient buffer pair i %% %d: every line has left the 256 MiB Infinity Cache before its next use" % pairs}

    pixels = float(h) * float(w)
    achieved = BYTES_PER_PIXEL * pixels / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and (h, w, q) == (4096, 4096, 50):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "%s (carried: PMC passes of an earlier rocprofv3 run of this command, tools/gpu_run.sh pmc_rd pmc_wr; not measured in this run)" % tj.get("source")
        except Exception:  # noqa: BLE001
            traffic = None
    out = {
        "metric": "Mpixels/s encode (DCT+quant kernel)",
        "value": round(pixels / (kernel_ms * 1e-3) / 1e6, 1),
        "unit": "Mpix/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(kernel_ms, 6),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32+f64" if args.variant == "hybrid" else "f64",
        "data": "synthetic",
        "config": {
            "workload": "single %dx%d random uint8 grayscale frame, quality=%d (%s), input resident in HBM"
            % (h, w, q, "BASELINE config 2" if (h, w) == (4096, 4096) else ("BASELINE config 5" if (h, w) == (16384, 16384) else "not a BASELINE size")),
            "kernel": args.variant,
            "frames_per_step_per_gpu": 1,
            "settle_ms": args.settle_ms,
            "untimed_launches": untimed + 1,
            "untimed_note": "back-to-back launches of the same kernel before the W warm-up steps (clock settling) + 1 statistics launch",
            "fallback_blocks_per_launch": fb.value,
            "parity": parity,
            "device": ctx.arch,
            "timed_region": "the last %d settling launches, the W warm-up launches, HIP event, the K timed launches, HIP event - one submission on the "
            "launch stream, inside the barrier + device-sync bracket, so that the first event is stamped when the last warm-up launch retires "
            "with the timed launches queued behind it: `value`, `ms_per_step` and `roofline.achieved` are all the interval between the two events "
            "(value x 3 B = roofline.achieved).  The host clock around the bracket (`wall_ms_per_step`) holds the untimed launches too, plus one "
            "submission ramp and one completion wake-up per CALL (`wall_overhead_us_total`)" % pre,
            "per_launch_us": per_launch,
            "launches_in_the_timed_submission": {"settling": pre, "warmup": args.warmup, "timed": args.steps},
            "wall_ms_per_step": round(wall_s * 1e3 / (args.steps + args.warmup + pre), 6),
            "wall_overhead_us_total": round((wall_s * 1e3 - kernel_ms * (args.steps + args.warmup + pre)) * 1e3, 1),
            "value_by_wall_clock": round(pixels * (args.steps + args.warmup + pre) / wall_s / 1e6, 1),
            "wall_note": "host clock over all launches of the bracket / their number; the overhead is the bracket minus that number x ms_per_step",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel_us": round(kernel_ms * 1e3, 3),
            "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * pixels,
            "note": "one frame and one coefficient buffer replayed: the 50 MB working set stays in the 256 MiB Infinity Cache (see cold)"
            if 3 * h * w < (200 << 20) else "one frame and one coefficient buffer replayed; %d MB per launch do not fit the 256 MiB Infinity Cache" % ((3 * h * w) >> 20),
            "limiter": "priced against HBM as the contract asks.  The kernel moves 1.005 x the algorithmic bytes; its loop alone (a timing build without any rare "
            "path) runs a 4096^2 launch in 7.7-7.8 us = 0.81 on every content; what the shipped kernel adds is the rare work on the launch's tail - the batch of "
            "blocks with an irrational coefficient inside its guard band (528 of 262,144 here: ~1.0 us, the launch ends with the unluckiest of 6,144 waves) and "
            "the strips whose rational ties are settled in the loop (one strip in six: ~0.3-1.0 us by box) - profiles/r05_content_ablation.txt, DESIGN.md 5.5",
        },
    }
    if cold is not None:
        out["roofline"]["cold"] = cold
        out["roofline"]["frac_hbm_cold"] = cold["frac"]  # the HBM figure: the warm one above is (partly) an Infinity-Cache figure
    for k in range(pairs):
        ctx.check(L.tic_dev_free(ctx.handle, d_imgs[k]))
        ctx.check(L.tic_dev_free(ctx.handle, d_outs[k]))
    return out


def shard_measurements(args, ctx, L, N, T, q, variant, first, count, ms, steps, warmup=10, barrier=None, timed=False):
    """BASELINE config 4, this rank's shard: frames first .. first+count-1 (1080x1920, seed 1234 + i).
    Kernel-only: the shard resident in HBM, one batched launch per step.  Host -> host: tic_compress_batch on the same frames."""
    h, w = 1080, 1920
    frames = [rand_frame(1234 + first + i, h, w) for i in range(count)]
    nblk = L.tic_num_blocks(h, w)
    img_bytes, coef_bytes = h * w, nblk * 128
    d_in, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img_bytes * count, C.byref(d_in)))
    ctx.check(L.tic_dev_alloc(ctx.handle, coef_bytes * count, C.byref(d_out)))
    for i, f in enumerate(frames):
        ctx.check(L.tic_memcpy_h2d(ctx.handle, C.c_void_p(d_in.value + i * img_bytes), f.ctypes.data, img_bytes))

    def launch(k):
        ctx.check(L.tic_dctq_dev_frames_timed(ctx.handle, d_in, count, h, w, w, img_bytes, q, d_out, coef_bytes, variant, k, C.byref(ms)))
        return k

    untimed = settle(args, ctx, L, lambda: launch(8))
    if warmup > 0:
        launch(warmup)
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    launch(steps)
    if barrier is not None:
        barrier()
    t_wall = time.perf_counter() - t0
    kernel_ms = ms.value / steps
    ctx.check(L.tic_dev_free(ctx.handle, d_in))
    ctx.check(L.tic_dev_free(ctx.handle, d_out))
    # whole pipeline host -> host through the C-ABI (caller-owned worst-case sized output buffers, as a C caller would hold
    # them; the second pass finds its staging and landing buffers faulted in)
    cap = L.tic_compress_bound(h, w)
    pool = np.empty((count, cap), dtype=np.uint8)
    inp = (C.c_void_p * count)(*[f.ctypes.data for f in frames])
    outp = (C.c_void_p * count)(*[pool[i].ctypes.data for i in range(count)])
    caps = (C.c_size_t * count)(*([cap] * count))
    lens = (C.c_size_t * count)()
    def h2h(inputs, lens_out, reps=5):
        """median of `reps` calls after one untimed call (the first call of a shape allocates the slots and faults them in), with
        the phase times of that call (tic_last_batch_phases)"""
        runs = []
        tr = (C.c_double * 8)()
        for r in range(reps + 1):
            t1 = time.perf_counter()
            ctx.check(L.tic_compress_batch(ctx.handle, inputs, count, h, w, w, q, outp, caps, lens_out, 0))
            dt = time.perf_counter() - t1
            ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
            if r:
                runs.append((dt, {"stage_or_register": round(tr[0], 2), "enqueue": round(tr[1], 2), "slot_wait": round(tr[5], 2),
                                  "chunk_wait": round(tr[2], 2), "read_back": round(tr[3], 2), "hand_out": round(tr[4], 2),
                                  "min_max_ms": None}))
        runs.sort(key=lambda x: x[0])
        dt, phases = runs[len(runs) // 2]
        phases["min_max_ms"] = [round(runs[0][0] * 1e3, 2), round(runs[-1][0] * 1e3, 2)]
        return dt, phases

    # pageable frames staged through the pipeline's pinned slots by copy threads (rounds 1-3's only route for pageable input) ...
    ctx.check(L.tic_set_auto_register(ctx.handle, 0))
    t_h2h_staged, ph_staged = h2h(inp, lens)
    # ... and pinned in place for the duration of the call (the default): one registration over the batch's address range
    ctx.check(L.tic_set_auto_register(ctx.handle, 1))
    t_h2h, ph = h2h(inp, lens)
    n_auto = C.c_int()
    ctx.check(L.tic_last_batch_auto_registered(ctx.handle, C.byref(n_auto)))
    sizes = [int(lens[i]) for i in range(count)]
    parity = manifest_parity(q, first, sizes, [pool[i, : sizes[i]] for i in range(count)])
    # the same with the caller's frames pinned (tic_host_register): no staging copy on the host, the H2D engine reads the frames
    # where they lie.  One registered block holding the shard, as a capture or decode buffer would be.
    block = np.empty((count, h, w), dtype=np.uint8)
    for i, f in enumerate(frames):
        block[i] = f
    ctx.check(L.tic_host_register(ctx.handle, block.ctypes.data, block.nbytes))
    inp_r = (C.c_void_p * count)(*[block[i].ctypes.data for i in range(count)])
    lens_r = (C.c_size_t * count)()
    t_h2h_reg, ph_reg = h2h(inp_r, lens_r)
    n_direct, n_staged = C.c_int(), C.c_int()
    ctx.check(L.tic_last_batch_input_path(ctx.handle, C.byref(n_direct), C.byref(n_staged)))
    ctx.check(L.tic_host_unregister(ctx.handle, block.ctypes.data))
    assert [int(lens_r[i]) for i in range(count)] == sizes, "registered-input pass produced different streams"
    node, ncpus = C.c_int(), C.c_int()
    ctx.check(L.tic_numa_info(ctx.handle, C.byref(node), C.byref(ncpus)))
    pixels = float(h) * w * count
    info = {
        "workload": "%d random 1920x1080 frames (seeds %d..%d), quality=%d" % (count, 1234 + first, 1234 + first + count - 1, q),
        "kernel_only_mpix_s": round(pixels / (kernel_ms * 1e-3) / 1e6, 1),
        "kernel_only_us_per_launch": round(kernel_ms * 1e3, 2),
        "kernel_only_frac_of_8TBs": round(BYTES_PER_PIXEL * pixels / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "host_to_host_mpix_s": round(pixels / t_h2h / 1e6, 1),
        "host_to_host_s": round(t_h2h, 5),
        "host_to_host_frames_per_s": round(count / t_h2h, 1),
        "host_to_host_phases_ms": ph,
        "host_to_host_note": "pageable caller frames (256 separate arrays), pinned in place for the call by ONE hipHostRegister over their address "
        "range (%d of %d frames took that route), median of 5 calls; phases are per pipeline thread and overlap.  The streams come back through "
        "a kernel that stores them into pinned host memory while the DMA engines upload (as DMA copies they queued behind four chunks of "
        "uploads: 13.8-14.2 ms, profiles/r04_batch_timeline.txt); the link's own bound for this batch is 9.5 ms" % (n_auto.value, count),
        "host_to_host_staged_s": round(t_h2h_staged, 5),
        "host_to_host_staged_frames_per_s": round(count / t_h2h_staged, 1),
        "host_to_host_staged_phases_ms": ph_staged,
        "host_to_host_staged_note": "the same frames copied into the pipeline's pinned slots by 8 host threads (tic_set_auto_register(0): "
        "rounds 1-3's route for pageable input: as fast as the pinned route on a quiet host, up to twice as slow with other tenants on it - see min_max_ms)",
        "frames_numa_nodes": frame_nodes(frames),
        "host_to_host_registered_mpix_s": round(pixels / t_h2h_reg / 1e6, 1),
        "host_to_host_registered_s": round(t_h2h_reg, 5),
        "host_to_host_registered_frames_per_s": round(count / t_h2h_reg, 1),
        "registered_frames_copied_in_place": int(n_direct.value),
        "numa": {"device_node": int(node.value), "cpus_of_that_node_in_this_process": int(ncpus.value),
                 "note": "the pipeline's own threads bind to those CPUs; pinned slots are placed on the device's node by hipHostMalloc"},
        "stream_bytes_total": int(sum(sizes)),
        "sizes_sha256": hashlib.sha256(np.asarray(sizes, dtype="<i8").tobytes()).hexdigest(),
        "parity": parity,
        "untimed_launches": untimed + warmup,
    }
    if timed:
        return info, t_wall, kernel_ms, sizes
    return info, sizes


def frame_nodes(frames):
    """NUMA nodes that hold the caller's frames (move_pages(2) in query mode on a sample of their pages); None where unavailable."""
    try:
        libc = C.CDLL(None, use_errno=True)
        sample = frames[:: max(1, len(frames) // 16)]
        pages = (C.c_void_p * len(sample))(*[(f.ctypes.data + f.nbytes // 2) & ~4095 for f in sample])
        status = (C.c_int * len(sample))()
        if libc.syscall(279, 0, C.c_ulong(len(sample)), pages, None, status, 0) != 0:
            return None
        return sorted(set(int(v) for v in status))
    except Exception:  # noqa: BLE001
        return None


def coefficient_parity(ctx, L, d_out, h, w, q, nblk):
    """The bench line carries its own proof: the coefficient buffer the TIMED launches wrote is read back and its dc / ac digests are
    compared with tests/golden/manifest.json (`rand1234_<h>x<w>_q<q>`: sha256 of the reference's own encode() output, dc differenced
    as codec.py:34-35, tests/golden/gen/make_goldens.py).  Raises on a difference: a benchmark of wrong output is not a benchmark."""
    path = os.path.join(ROOT, "tests", "golden", "manifest.json")
    key = "rand1234_%dx%d_q%d" % (h, w, q)
    if not os.path.exists(path):
        return {"status": "unchecked", "why": "tests/golden/manifest.json missing"}
    ent = json.load(open(path))["entries"].get(key)
    if ent is None or "dc_i4_sha256" not in ent:
        return {"status": "unchecked", "why": "no manifest entry %s" % key}
    zz = np.empty((nblk, 64), dtype=np.int16)
    ctx.check(L.tic_memcpy_d2h(ctx.handle, zz.ctypes.data, d_out, nblk * 128))
    dc = zz[:, 0].astype(np.int32)
    dc[1:] = np.diff(zz[:, 0].astype(np.int32))
    got_dc = hashlib.sha256(dc.astype("<i4").tobytes()).hexdigest()
    got_ac = hashlib.sha256(np.ascontiguousarray(zz[:, 1:]).astype("<i4").tobytes()).hexdigest()
    if got_dc != ent["dc_i4_sha256"] or got_ac != ent["ac_i4_sha256"]:
        raise AssertionError("coefficients of the timed launches differ from the reference's (%s): dc %s ac %s" % (key, got_dc[:16], got_ac[:16]))
    return {"status": "ok", "against": "tests/golden/manifest.json %s (sha256 of the reference's encode() dc / ac)" % key,
            "dc_i4_sha256": got_dc, "ac_i4_sha256": got_ac, "checked": "the coefficient buffer written by the timed launches"}


def manifest_parity(q, first, sizes, streams):
    """Full-size parity of the config-3/4 shard: every stream tic_compress_batch returned against tests/golden/manifest_r4.json -
    size and sha256 per frame, produced by the UNMODIFIED reference's compress() (codec.py:133-164) on the same seeded frames
    (tests/golden/gen/make_goldens_r4.py).  Raises on any difference: a benchmark of wrong output is not a benchmark."""
    path = os.path.join(ROOT, "tests", "golden", "manifest_r4.json")
    if q != 50 or not os.path.exists(path):
        return {"status": "unchecked", "why": "no manifest for this quality" if q != 50 else "tests/golden/manifest_r4.json missing"}
    m = json.load(open(path))
    by_seed = {f["seed"]: f for f in m["frames"]}
    checked, sources = 0, set()
    for i, (n, st) in enumerate(zip(sizes, streams)):
        f = by_seed.get(1234 + first + i)
        if f is None:
            continue
        digest = hashlib.sha256(np.ascontiguousarray(st).tobytes()).hexdigest()
        if n != f["bytes"] or digest != f["sha256"]:
            raise AssertionError("frame %d (seed %d): stream of %d bytes sha256 %s, the reference's has %d bytes sha256 %s"
                                 % (first + i, 1234 + first + i, n, digest[:16], f["bytes"], f["sha256"][:16]))
        checked += 1
        sources.add(f.get("source", "reference"))
    out = {"status": "ok" if checked == len(sizes) else ("partial" if checked else "unchecked"), "frames_checked": checked, "frames": len(sizes),
           "against": "tests/golden/manifest_r4.json (per-frame size + sha256 of the %s's compress())" % " / ".join(sorted(sources) or ["reference"])}
    if first == 0 and len(sizes) == 256 and checked == 256:
        out["sizes_sha256_matches_manifest"] = hashlib.sha256(np.asarray(sizes, dtype="<i8").tobytes()).hexdigest() == m.get("sizes_sha256_first256")
    return out


def codec_resident(args, ctx, L, N, q):
    """The callers either side of the hot path on the same frame, everything resident in HBM: the whole encoder
    (transform + device entropy stage, tic_compress_dev: the stream is byte-identical to the reference's, tests/) and the whole decoder
    (device Huffman decode + inverse transform, tic_decompress_dev).  Informational: never `value`."""
    import hashlib
    h, w = args.height, args.width
    img = rand_frame(1234, h, w)
    cap = L.tic_compress_bound(h, w)
    d_img, d_out, d_pix = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap + 64, C.byref(d_out)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_pix)))
    try:
        ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
        n = C.c_size_t()
        def timed(fn, reps=40):
            for _ in range(5): fn()
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(reps): fn()
                t.append((time.perf_counter() - t0) / reps)
            return sorted(t)[len(t) // 2]
        t_enc = timed(lambda: ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, q, d_out, cap, C.byref(n))))
        stream = np.empty(n.value, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, stream.ctypes.data, d_out, n.value))
        # the same encoder, 64 resident frames back to back through the asynchronous form: one submission ramp and one wake-up per burst
        burst = 64
        tickets = [C.c_longlong() for _ in range(burst)]
        def burst_fn():
            for k in range(burst):
                ctx.check(L.tic_compress_dev_async(ctx.handle, d_img, h, w, w, q, d_out, cap, C.byref(tickets[k])))
            for k in range(burst):
                nn = C.c_size_t()
                ctx.check(L.tic_async_result(ctx.handle, tickets[k].value, 1, C.byref(nn)))
                assert nn.value == n.value, (nn.value, n.value)
        t_pipe = timed(burst_fn, reps=3) / burst
        stream2 = np.empty(n.value, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, stream2.ctypes.data, d_out, n.value))
        assert np.array_equal(stream, stream2), "the asynchronous form wrote a different stream"
        # ... and with a SECOND context on the same device doing the same from a thread of its own: two streams, whose kernels overlap (the
        # transform is bound by HBM, the pack kernel by vector issue); frames of both contexts per second
        t_pipe2 = None
        try:
            ctx_b = type(ctx)(ctx.device) if hasattr(ctx, "device") else None
        except Exception:  # noqa: BLE001
            ctx_b = None
        if ctx_b is not None:
            d_img_b, d_out_b = C.c_void_p(), C.c_void_p()
            try:
                ctx_b.check(L.tic_dev_alloc(ctx_b.handle, img.size, C.byref(d_img_b)))
                ctx_b.check(L.tic_dev_alloc(ctx_b.handle, cap + 64, C.byref(d_out_b)))
                ctx_b.check(L.tic_memcpy_h2d(ctx_b.handle, d_img_b, img.ctypes.data, img.size))
                def burst_on(c, di, do, reps):
                    tk = [C.c_longlong() for _ in range(burst)]
                    for _ in range(reps):
                        for k in range(burst):
                            c.check(L.tic_compress_dev_async(c.handle, di, h, w, w, q, do, cap, C.byref(tk[k])))
                        for k in range(burst):
                            nn = C.c_size_t()
                            c.check(L.tic_async_result(c.handle, tk[k].value, 1, C.byref(nn)))
                burst_on(ctx_b, d_img_b, d_out_b, 2)
                samples = []
                for _ in range(5):
                    th = [threading.Thread(target=burst_on, args=(ctx, d_img, d_out, 4)), threading.Thread(target=burst_on, args=(ctx_b, d_img_b, d_out_b, 4))]
                    t0 = time.perf_counter()
                    for t in th: t.start()
                    for t in th: t.join()
                    samples.append((time.perf_counter() - t0) / (2 * 4 * burst))
                t_pipe2 = sorted(samples)[len(samples) // 2]
                sb = np.empty(n.value, np.uint8)
                ctx_b.check(L.tic_memcpy_d2h(ctx_b.handle, sb.ctypes.data, d_out_b, n.value))
                assert np.array_equal(stream, sb), "the second context wrote a different stream"
            finally:
                for p in (d_img_b, d_out_b):
                    if p.value: L.tic_dev_free(ctx_b.handle, p)
                ctx_b.close()
        t_dec = timed(lambda: ctx.check(L.tic_decompress_dev(ctx.handle, d_out, n.value, d_pix, w, img.size, None, None)))
        back_sync = np.empty((h, w), np.uint8)
        ctx.check(L.tic_set_decode_guess(ctx.handle, 0))  # the same call reading the 16-byte header from device memory first (no launch on a guess)
        t_dec_read = timed(lambda: ctx.check(L.tic_decompress_dev(ctx.handle, d_out, n.value, d_pix, w, img.size, None, None)))
        ctx.check(L.tic_set_decode_guess(ctx.handle, 1))
        for _ in range(3):
            ctx.check(L.tic_decompress_dev(ctx.handle, d_out, n.value, d_pix, w, img.size, None, None))
        rb, tr = C.c_int(), C.c_int()
        ctx.check(L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr)))
        guess_held = int(L.tic_last_decode_guess(ctx.handle))
        # the same decoder with four tickets open (tic_decompress_dev_async): 64 frames, a frame is collected when the fourth behind it has been queued
        dec_burst, dec_open = 64, 4
        d_pix4 = []
        for k in range(dec_open):
            p = C.c_void_p()
            ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(p)))
            d_pix4.append(p)
        try:
            def dec_burst_fn():
                tk = []
                for k in range(dec_burst):
                    if len(tk) == dec_open:
                        ctx.check(L.tic_decompress_async_result(ctx.handle, tk.pop(0), 1, None, None))
                    t = C.c_longlong()
                    ctx.check(L.tic_decompress_dev_async(ctx.handle, d_out, n.value, d_pix4[k % dec_open], w, img.size, C.byref(t)))
                    tk.append(t.value)
                while tk:
                    ctx.check(L.tic_decompress_async_result(ctx.handle, tk.pop(0), 1, None, None))
            t_dec_pipe = timed(dec_burst_fn, reps=3) / dec_burst
            for p in d_pix4:  # every destination holds the same pixels as the synchronous call's
                b4 = np.empty((h, w), np.uint8)
                ctx.check(L.tic_memcpy_d2h(ctx.handle, b4.ctypes.data, p, b4.size))
                ctx.check(L.tic_memcpy_d2h(ctx.handle, back_sync.ctypes.data, d_pix, back_sync.size))
                assert np.array_equal(b4, back_sync), "the asynchronous decode wrote different pixels"
        finally:
            for p in d_pix4:
                L.tic_dev_free(ctx.handle, p)
        back = np.empty((h, w), np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, back.ctypes.data, d_pix, back.size))
        err = np.abs(back.astype(np.int16) - img.astype(np.int16))
        # the leg proves itself where a fixture exists: stream and decoded pixels against the digests the pinned oracle left for this frame
        # (tests/golden/big_frame_decode.json, written by tests/golden/gen/make_goldens_r5.py in the build container - data, not the oracle)
        stream_sha, pixel_sha = hashlib.sha256(stream.tobytes()).hexdigest(), hashlib.sha256(back.tobytes()).hexdigest()
        parity = {"status": "no fixture for this frame"}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "big_frame_decode.json")) as f:
                for e in json.load(f)["entries"]:
                    if (e["seed"], e["height"], e["width"], e["quality"]) == (1234, h, w, q):
                        ok = e["sha256"] == stream_sha and e["decoded_sha256"] == pixel_sha and e["bytes"] == n.value
                        if not ok:
                            raise AssertionError("codec_resident: stream %s / pixels %s differ from the oracle's digests" % (stream_sha[:16], pixel_sha[:16]))
                        parity = {"status": "ok", "against": "tests/golden/big_frame_decode.json (sha256 of the pinned oracle's stream and decoded pixels for this frame)",
                                  "checked": "the stream tic_compress_dev wrote and the pixels tic_decompress_dev decoded from it in the timed calls"}
        except OSError as ex:
            parity = {"status": "fixture unreadable: %s" % ex}
        return {
            "parity": parity,
            "workload": "the %dx%d frame of config 2 (seed 1234), quality=%d, image, stream and pixels resident in HBM" % (h, w, q),
            "compress_dev_us": round(t_enc * 1e6, 1), "compress_dev_mpix_s": round(h * w / t_enc / 1e6, 1),
            "pipelined_us": round(t_pipe * 1e6, 1), "pipelined_mpix_s": round(h * w / t_pipe / 1e6, 1),
            "pipelined_note": "tic_compress_dev_async: %d frames queued back to back, results collected afterwards (same stream bytes); per frame" % burst,
            "pipelined_two_contexts_us": (round(t_pipe2 * 1e6, 1) if t_pipe2 else None),
            "pipelined_two_contexts_note": "the same from two contexts on this device, a host thread each (two HIP streams): wall time / frames of both",
            "stream_bytes": int(n.value), "stream_sha256": stream_sha, "decoded_sha256": pixel_sha,
            "decompress_dev_us": round(t_dec * 1e6, 1), "decompress_dev_mpix_s": round(h * w / t_dec / 1e6, 1),
            "decoder_path": int(L.tic_last_decode_path(ctx.handle)), "decoder_range_bits": rb.value, "decoder_runs": tr.value,
            "decoder_header_guess": guess_held, "decompress_dev_header_read_first_us": round(t_dec_read * 1e6, 1),
            "decompress_pipelined_us": round(t_dec_pipe * 1e6, 1), "decompress_pipelined_mpix_s": round(h * w / t_dec_pipe / 1e6, 1),
            "decompress_pipelined_note": "tic_decompress_dev_async: %d frames, %d tickets open at a time on streams of their own (same pixels); per frame" % (dec_burst, dec_open),
            "decoder_note": "tic_decompress_dev launches on a guess of the stream's header (the header of the stream this context decoded last, once two streams in a row "
                            "came with the same one; 1 = the guess held, as for every call of this loop but the first two) instead of reading 16 bytes from device memory first; "
                            "a wrong guess costs a second decode; decompress_dev_header_read_first_us is the same call with the guessing switched off (tic_set_decode_guess)",
            "round_trip_max_abs_error": int(err.max()), "round_trip_mean_abs_error": round(float(err.mean()), 3),
            "note": "host clock around the C-ABI call (launches + one wait included), median of 5 x 40 calls; parity of both directions is the test suite's business "
                    "(streams against the reference's, pixels against the reference's decoder) - the round-trip error here is the quantiser's",
        }
    finally:
        for p in (d_img, d_out, d_pix):
            L.tic_dev_free(ctx.handle, p)


def bench_set(args, ctx, L):
    """The reference's own benchmark workload (/root/reference/tests/benchmark.py:12-23: 49 images of 512 x 512, quality 90, 80, 50, 20, 10, 5; per image
    compress() then decompress()) as TWO calls per quality, host memory to host memory: tic_compress_batch (49 images -> 49 streams) and
    tic_decompress_batch (49 streams -> 49 images).  Pixels and the expected size / sha256 of every stream and decoded image are the
    committed fixtures tests/golden/benchmark_set.{npz,json}, produced by the unmodified reference (tests/golden/gen/make_goldens_r5.py);
    every stream and every image of the timed calls is checked against them.  Informational: never `value`."""
    gold = os.path.join(ROOT, "tests", "golden")
    try:
        man = json.load(open(os.path.join(gold, "benchmark_set.json")))["entries"]
        px = np.load(os.path.join(gold, "benchmark_set.npz"))["pixels"]
    except OSError as ex:
        return {"status": "fixture unreadable: %s" % ex}
    by = {(e["image"], e["quality"]): e for e in man}
    n, h, w = px.shape
    frames = [np.ascontiguousarray(px[i]) for i in range(n)]
    cap = L.tic_compress_bound(h, w)
    pool = np.empty((n, cap), dtype=np.uint8)
    inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
    caps = (C.c_size_t * n)(*([cap] * n))
    lens = (C.c_size_t * n)()
    block = np.empty((n, h, w), dtype=np.uint8)
    pix_p = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)])
    pix_c = (C.c_size_t * n)(*([h * w] * n))
    rows, checked = {}, 0
    tot_c = tot_d = 0.0
    per_call = None
    for q in (90, 80, 50, 20, 10, 5):
        def comp():
            ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        comp()
        tc = []
        for _ in range(5):
            t0 = time.perf_counter(); comp(); tc.append(time.perf_counter() - t0)
        streams = [pool[i, : lens[i]].copy() for i in range(n)]
        sp = (C.c_void_p * n)(*[s_.ctypes.data for s_ in streams])
        sl = (C.c_size_t * n)(*[s_.size for s_ in streams])
        def dec():
            ctx.check(L.tic_decompress_batch(ctx.handle, sp, sl, n, pix_p, pix_c, None, None))
        dec()
        td = []
        for _ in range(5):
            block[:] = 0
            t0 = time.perf_counter(); dec(); td.append(time.perf_counter() - t0)
        nb, ns, nc, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ctx.check(L.tic_last_decompress_batch(ctx.handle, C.byref(nb), C.byref(ns), C.byref(nc), C.byref(nd)))
        for i in range(n):
            e = by[(i + 1, q)]
            if streams[i].size != e["bytes"] or hashlib.sha256(streams[i].tobytes()).hexdigest() != e["sha256"]:
                raise AssertionError("bench_set: stream of image %d at q = %d differs from the reference's" % (i + 1, q))
            if hashlib.sha256(block[i].tobytes()).hexdigest() != e["decoded_sha256"]:
                raise AssertionError("bench_set: decoded image %d at q = %d differs from the reference's" % (i + 1, q))
            checked += 1
        c, d = sorted(tc)[2], sorted(td)[2]
        tot_c += c; tot_d += d
        rows["q%d" % q] = {"compress_batch_ms": round(c * 1e3, 3), "compress_images_per_s": round(n / c, 0), "decompress_batch_ms": round(d * 1e3, 3),
                           "decompress_images_per_s": round(n / d, 0), "stream_kb_mean": round(float(sum(int(lens[i]) for i in range(n))) / n / 1024, 1),
                           "decode": {"batch_frames": nb.value, "single_frames": ns.value, "chunks": nc.value, "copied_in_place": nd.value}}
        if q == 50:  # the same 49 pairs one call per image, as the reference's loop is written (C-ABI: tic_compress / tic_decompress)
            o1, n1 = np.empty(cap, np.uint8), C.c_size_t()
            p1 = np.empty((h, w), np.uint8)
            for i in range(n):  # one untimed pass: the first call of a geometry allocates the context's small-frame buffers, and the frames were
                # pinned in place and released again by the batch call in front (their first upload as pageable memory afterwards costs 0.2 ms each)
                ctx.check(L.tic_compress(ctx.handle, frames[i].ctypes.data, h, w, w, q, o1.ctypes.data, cap, C.byref(n1)))
                ctx.check(L.tic_decompress(ctx.handle, streams[i].ctypes.data, streams[i].size, p1.ctypes.data, p1.size))
            t0 = time.perf_counter()
            for i in range(n):
                ctx.check(L.tic_compress(ctx.handle, frames[i].ctypes.data, h, w, w, q, o1.ctypes.data, cap, C.byref(n1)))
            t1 = time.perf_counter()
            for i in range(n):
                ctx.check(L.tic_decompress(ctx.handle, streams[i].ctypes.data, streams[i].size, p1.ctypes.data, p1.size))
            t2 = time.perf_counter()
            per_call = {"quality": 50, "compress_us": round((t1 - t0) / n * 1e6, 1), "decompress_us": round((t2 - t1) / n * 1e6, 1),
                        "compress_images_per_s": round(n / (t1 - t0), 0), "decompress_images_per_s": round(n / (t2 - t1), 0)}
    return {"workload": "%d images of %dx%d (the reference's data/1..49.gif as committed pixels) x quality 90, 80, 50, 20, 10, 5: tic_compress_batch then "
                        "tic_decompress_batch, host memory to host memory, median of 5 calls each" % (n, h, w),
            "compress_images_per_s": round(6 * n / tot_c, 0), "decompress_images_per_s": round(6 * n / tot_d, 0), "by_quality": rows, "per_call": per_call,
            "parity": {"status": "ok", "pairs_checked": checked, "against": "tests/golden/benchmark_set.json (size + sha256 of the reference's stream and of its "
                       "decoded image, per image and quality)", "checked": "the streams and images the timed calls produced"},
            "reference": "the reference's compress() takes ~0.3 s and its decompress() ~0.14 s per 512 x 512 image on its authors' laptop (BASELINE.md section 1)"}


def cpu_baseline(img, q, budget_s, threads):
    """The oracle (C restatement of the reference's CPU path for this stage; bit-identical output) timed on this host.
    threads = 1: whole passes over the frame on one thread.  threads = 0: this process's CPU share, at most 16 threads (the GPU box's
    share per GPU); threads = -1: every CPU this process may run on ("all host cores").  With more than one thread the frame
    is cut into bands of block rows, one band per thread, and every thread runs its band `reps` times in a row (ctypes releases the
    GIL; starting 256 Python threads per pass would cost more than the pass)."""
    from oracle import pyoracle

    pyoracle.build()
    h, w = img.shape
    pyoracle.encode_zz16(img[:64], q)  # warm up
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    T = 1 if threads == 1 else max(1, min(avail, 16) if threads == 0 else avail)
    rows = (h + 7) // 8
    T = min(T, rows)
    bands = [(rows * t // T * 8, min(h, rows * (t + 1) // T * 8)) for t in range(T)]

    def passes(reps):
        if T == 1:
            for _ in range(reps):
                pyoracle.encode_zz16(img, q)
            return

        def work(a, b):
            band = img[a:b]
            for _ in range(reps):
                pyoracle.encode_zz16(band, q)

        th = [threading.Thread(target=work, args=(a, b)) for a, b in bands if b > a]
        for t in th:
            t.start()
        for t in th:
            t.join()

    passes(1)  # untimed: the threads' first touch of their bands
    n, reps, t0 = 0, 1, time.perf_counter()
    while True:
        t1 = time.perf_counter()
        passes(reps)
        n += reps
        now = time.perf_counter()
        dt = now - t0
        if dt >= budget_s or n >= 4096:
            break
        per = (now - t1) / reps
        reps = max(1, min(512, int(min(budget_s - dt, budget_s / 3) / max(per, 1e-6))))  # a few long rounds instead of many short ones
    return {
        "value": round(n * h * w / dt / 1e6, 2),
        "unit": "Mpix/s",
        "cores": T,
        "kind": "port",
        "sample": "%d full passes of the same %dx%d frame, q=%d, transform stage only (oracle tico_encode_zz16%s), %.1f s on %d of %d host "
        "threads (%s); the reference's own numpy/scipy encode() measured 21.4 Mpix/s on one thread in the build container (BASELINE.md)"
        % (n, h, w, q, "" if T == 1 else ", bands of block rows, every thread its band", dt, T, os.cpu_count() or 0, cpu_model()),
    }


def cpu_baseline_numpy_scipy(img, q, budget_s):
    """the like-for-like "pure-Python-stack" figure - encode() restated on numpy + scipy.fftpack (oracle/np_encode.py,
    equal to the C oracle coefficient for coefficient), single thread as the reference runs it.  None where scipy is missing."""
    try:
        import scipy  # noqa: F401

        from oracle import np_encode
    except ImportError:
        return None
    h, w = img.shape
    np_encode.encode(img[:64], q)
    n, t0 = 0, time.perf_counter()
    while True:
        np_encode.encode(img, q)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 64:
            break
    return {"value": round(n * h * w / dt / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "%d full passes of the same %dx%d frame, q=%d, oracle/np_encode.py (numpy %s + scipy %s: pad, level shift, "
            "scipy.fftpack.dct twice, np.round(X / div), zig-zag gather, DC difference), %.1f s on one thread (%s); the reference's own "
            "encode() measured 21.4 Mpix/s in the build container (BASELINE.md)" % (n, h, w, q, np.__version__, scipy.__version__, dt, cpu_model())}


if __name__ == "__main__":
    main()
