#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X tinyimgcodec hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the transform stage (level shift -> 2-D DCT -> quantise -> zig-zag, i.e. the reference's
encode(), codec.py:26-43) over one synthetic frame per GPU, input already resident in HBM.  At N = 1 the workload is
BASELINE config 2: one 4096x4096 random uint8 frame (numpy default_rng(1234)), quality 50.  At N > 1 every rank
runs the same per-GPU workload on its own frame (independent frames, no data-path collective: weak scaling); RCCL is
used only where the north star has it - an all-gather of per-frame compressed sizes, outside the timed region.

Rank 0 prints ONE JSON line.  `value` = whole-job Mpixel/s = (frames x pixels x K) / max-over-ranks wall time.
`roofline.achieved` = algorithmic bytes (3 B/pixel: 1 B read + 2 B written) / average kernel duration measured
with HIP events recorded on the library's own stream around the same K launches.  `cpu_baseline` = the oracle (C
restatement of the reference's CPU path, single thread) timed on this host on the same frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
BYTES_PER_PIXEL = 3.0  # 1 B uint8 read + 2 B int16 written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--quality", type=int, default=50)
    ap.add_argument("--variant", choices=["hybrid", "exact"], default="hybrid")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=60.0, help="untimed back-to-back launches before the warm-up steps (clock settling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    dist = None
    torch = None
    # TIC_BENCH_BACKEND=gloo + TIC_BENCH_SHARE_GPU=1 rehearse the multi-rank flow on a one-GPU box (all ranks on
    # device 0, CPU tensors for the collectives); the real run uses RCCL with one rank per GPU.
    backend = os.environ.get("TIC_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("TIC_BENCH_SHARE_GPU", "0") == "1"
    tdev = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist

        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            tdev = "cuda"
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    import tinyimgcodec_amd as T
    from tinyimgcodec_amd import _native as N

    L = N.load()
    ctx = T.Context(0 if share_gpu else local_rank)  # raises loudly if the HIP library / an MI355X is missing
    h, w, q = args.height, args.width, args.quality
    variant = N.KERNEL_HYBRID if args.variant == "hybrid" else N.KERNEL_EXACT

    # synthetic frame of this rank (seed 1234 + rank), uploaded once: the timed region starts with data in HBM
    img = np.random.default_rng(1234 + rank).integers(0, 256, (h, w), dtype=np.uint8)
    pitch = (w + 255) // 256 * 256
    host = np.zeros((h, pitch), dtype=np.uint8)
    host[:, :w] = img
    nblk = L.tic_num_blocks(h, w)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, nblk * 128, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, host.ctypes.data, host.size))

    def barrier():
        ctx.check(L.tic_sync(ctx.handle))
        if dist is not None:
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()

    ms = C.c_float(0.0)
    # one untimed launch with the diagnostic counter on: how many blocks leave the fast path on this frame
    fb = C.c_ulonglong(0)
    ctx.check(L.tic_set_stats(ctx.handle, 1))
    ctx.check(L.tic_last_fallback_blocks(ctx.handle, C.byref(fb)))  # resets the counter
    ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, pitch, q, d_out, variant))
    ctx.check(L.tic_last_fallback_blocks(ctx.handle, C.byref(fb)))
    ctx.check(L.tic_set_stats(ctx.handle, 0))
    # clock settling (untimed, before the W warm-up steps): the chip needs a few ms of back-to-back launches to reach its
    # sustained clocks - 200 launches after 20 read 15.4 us where 5000 after 2000 read 12.1 us
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, variant, 256, C.byref(ms)))
    if args.warmup > 0:
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, variant, args.warmup, C.byref(ms)))
    barrier()
    t0 = time.perf_counter()
    # exactly K launches, bracketed by HIP events on the launch stream; returns after the stream has drained
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, variant, args.steps, C.byref(ms)))
    barrier()
    t1 = time.perf_counter()
    wall_s = t1 - t0
    kernel_ms = ms.value / args.steps

    sizes = None
    if dist is not None:
        tmax = torch.tensor([wall_s, kernel_ms], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall_s, kernel_ms_max = float(tmax[0]), float(tmax[1])
        # the north star's only collective (RCCL all-gather of per-frame compressed sizes), outside the timed
        # region: each rank entropy-codes a 512x512 crop of its frame on the host and the sizes are gathered
        from tinyimgcodec_amd.distributed import gather_sizes

        mine = [len(T.compress(img[:512, :512], q, ctx=ctx))]
        allsz, _ = gather_sizes(mine, world, device=tdev)
        sizes = [int(v) for v in allsz]
    else:
        kernel_ms_max = kernel_ms

    if rank == 0:
        pixels = float(h) * float(w)
        value = pixels * world * args.steps / wall_s / 1e6  # Mpixel/s, whole job
        achieved = BYTES_PER_PIXEL * pixels / (kernel_ms_max * 1e-3) / 1e9  # GB/s of one kernel launch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and (h, w, q) == (4096, 4096, 50):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "Mpixels/s encode (DCT+quant kernel)",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall_s * 1e3 / args.steps, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64" if args.variant == "hybrid" else "f64",
            "data": "synthetic",
            "config": {
                "workload": "single %dx%d random uint8 grayscale frame per GPU, quality=%d (BASELINE config 2), "
                "input resident in HBM" % (h, w, q),
                "kernel": args.variant,
                "frames_per_step_per_gpu": 1,
                "sharding": "independent frames, one per rank; no data-path collective",
                "fallback_blocks_per_launch": fb.value,
                "device": ctx.arch,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel_us": round(kernel_ms_max * 1e3, 3),
                "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * pixels,
            },
        }
        if sizes is not None:
            out["config"]["rccl_gathered_sizes"] = sizes
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(img, q, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    ctx.check(L.tic_dev_free(ctx.handle, d_img))
    ctx.check(L.tic_dev_free(ctx.handle, d_out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(img, q, budget_s):
    """The oracle (C restatement of the reference's CPU path for this stage; bit-identical output, single thread)
    timed on this host.  Sample: whole passes over the same frame until ~budget_s seconds are spent."""
    from oracle import pyoracle

    pyoracle.build()
    h, w = img.shape
    pyoracle.encode_zz16(img[:64], q)  # warm up
    n, t0 = 0, time.perf_counter()
    while True:
        pyoracle.encode_zz16(img, q)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 64:
            break
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(n * h * w / dt / 1e6, 2),
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d full passes of the same %dx%d frame, q=%d, transform stage only (oracle tico_encode_zz16), "
        "%.1f s on 1 of %d host threads (%s); the reference's own numpy/scipy encode() measured 21.4 Mpix/s in the "
        "build container (BASELINE.md)" % (n, h, w, q, dt, os.cpu_count() or 0, cpu_model),
    }


if __name__ == "__main__":
    main()
