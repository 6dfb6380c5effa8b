"""A/B timing of kernel variants on COLD data in one process: 12 rotating frame/coefficient pairs (604 MB at 4096^2, more than the
256 MiB Infinity Cache holds), settings interleaved round-robin like tools/ab.py.  Usage: python tools/ab_cold.py [--variants 2,610]"""
import argparse, ctypes as C, os, statistics, sys
os.environ["TIC_TUNE"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
os.environ["TIC_TEST_HOOKS"] = "1"  # the test-hooks build of the product sources honours the schedule knobs
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="2,610")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=240)
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--pairs", type=int, default=12)
ap.add_argument("--quality", type=int, default=50)
ap.add_argument("settings", nargs="*", default=[""], help='knob settings to interleave, e.g. "TIC_SPLIT=16,13,10,7,4,2" ""')
args = ap.parse_args()
KNOBS = ("TIC_MAX_WGS", "TIC_SCHED", "TIC_CHUNK", "TIC_LDS_PAD", "TIC_SPLIT", "TIC_NOCAP", "TIC_ORDER")
def apply(setting):
    for k in KNOBS: os.environ.pop(k, None)
    for kv in setting.split():
        k, v = kv.split("="); os.environ[k] = v
L = N.load(); ctx = T.Context(0)
h = w = args.dim
imgs, outs = [], []
for k in range(args.pairs):
    img = np.random.default_rng(1234 + k).integers(0, 256, (h, w), dtype=np.uint8)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    imgs.append(d_img); outs.append(d_out)
d_imgs = (C.c_void_p * args.pairs)(*[p.value for p in imgs]); d_outs = (C.c_void_p * args.pairs)(*[p.value for p in outs])
variants = [int(v) for v in args.variants.split(",")]
ms = C.c_float()
res = {(s, v): [] for s in args.settings for v in variants}
for rnd in range(args.rounds + 2):
    for s in args.settings:
        apply(s)
        for v in variants:
            ctx.check(L.tic_dctq_dev_timed_rotating(ctx.handle, d_imgs, d_outs, args.pairs, h, w, w, args.quality, v, args.iters, C.byref(ms)))
            if rnd >= 2: res[(s, v)].append(ms.value * 1e3 / args.iters)
for s in args.settings:
    for v in variants:
        r = res[(s, v)]
        print("%5d^2 cold (%d pairs) q=%d %-28s variant %4d  median %7.2f us  min %7.2f  max %7.2f   %6.1f GB/s" %
              (args.dim, args.pairs, args.quality, s or "(defaults)", v, statistics.median(r), min(r), max(r), 3.0 * h * w / statistics.median(r) / 1e3), flush=True)
