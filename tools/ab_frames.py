"""A/B of the launcher's schedule knobs on BASELINE config 4's kernel-only leg: 256 resident 1920x1080 frames, one batched launch
(tic_dctq_dev_frames_timed), settings interleaved round-robin in one process.  Product library (the knobs are test hooks there).
Usage: python tools/ab_frames.py [--frames 256] "TIC_CHUNK=8" "TIC_CHUNK=32" ...   ('' = defaults)"""
import argparse, ctypes as C, os, statistics, sys
os.environ["TIC_TEST_HOOKS"] = "1"; os.environ["TIC_TUNE"] = "1"
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("settings", nargs="*", default=[""])
args = ap.parse_args()
L = N.load(); ctx = T.Context(0)
h, w, n = 1080, 1920, args.frames
img_bytes, coef_bytes = h * w, L.tic_num_blocks(h, w) * 128
d_in, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img_bytes * n, C.byref(d_in)))
ctx.check(L.tic_dev_alloc(ctx.handle, coef_bytes * n, C.byref(d_out)))
blk = np.random.default_rng(1).integers(0, 256, (16, h, w), dtype=np.uint8)
for i in range(0, n, 16):
    m = min(16, n - i)
    ctx.check(L.tic_memcpy_h2d(ctx.handle, C.c_void_p(d_in.value + i * img_bytes), blk.ctypes.data, img_bytes * m))
KNOBS = ("TIC_MAX_WGS", "TIC_SCHED", "TIC_CHUNK", "TIC_SPLIT")
ms = C.c_float()
res = {s: [] for s in args.settings}
for rnd in range(args.rounds + 1):
    for s in args.settings:
        for k in KNOBS: os.environ.pop(k, None)
        for kv in s.split():
            k, v = kv.split("="); os.environ[k] = v
        ctx.check(L.tic_dctq_dev_frames_timed(ctx.handle, d_in, n, h, w, w, img_bytes, 50, d_out, coef_bytes, 2, args.iters, C.byref(ms)))
        if rnd: res[s].append(ms.value * 1e3 / args.iters)
for s in args.settings:
    r = res[s]; med = statistics.median(r)
    print("%d x 1080p %-28s median %8.2f us  min %8.2f  max %8.2f   %6.1f GB/s (%.3f of 8 TB/s)" % (n, s or "(defaults)", med, min(r), max(r), 3.0 * h * w * n / med / 1e3, 3.0 * h * w * n / med / 1e3 / 8000))
