"""A/B timing of tuning-knob settings of the hybrid kernel in ONE process (TIC_TUNE makes the library re-read its knobs at
every launch): settings are interleaved round-robin, medians over the rounds are reported, so clock and thermal drift hits
all settings alike.  Usage: python tools/ab.py [--dims 4096,16384] [--rounds 7] "K=V K=V" "K=V" ...   ('' = defaults)"""
import argparse, ctypes as C, os, statistics, sys
os.environ["TIC_TUNE"] = "1"
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
os.environ["TIC_TEST_HOOKS"] = "1"  # the test-hooks build of the product sources honours the schedule knobs
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="4096,16384")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--variants", default="2,12,15")
ap.add_argument("--iters", type=int, default=0, help="launches per sample (default 50 at 4096^2, 8 above)")
ap.add_argument("settings", nargs="*", default=[""])
args = ap.parse_args()
L = N.load(); ctx = T.Context(0)
KNOBS = ("TIC_MAX_WGS", "TIC_SCHED", "TIC_CHUNK", "TIC_LDS_PAD", "TIC_SPLIT", "TIC_NOCAP")
names = {2: "full", 12: "no post-pass", 15: "skeleton", 10: "no arithmetic", 1: "exact", 18: "compute only", 19: "compute only, no LDS", 20: "compute only, no arithmetic", 40: "lane kernel", 41: "lane kernel, rare paths off"}
def apply(setting):
    for k in KNOBS: os.environ.pop(k, None)
    for kv in setting.split():
        k, v = kv.split("="); os.environ[k] = v
for dim in [int(x) for x in args.dims.split(",")]:
    h = w = dim
    img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    ms = C.c_float()
    iters = args.iters or (50 if dim <= 4096 else 8)
    variants = [int(v) for v in args.variants.split(",")]
    res = {(s, v): [] for s in args.settings for v in variants}
    for rnd in range(args.rounds + 1):
        for s in args.settings:
            apply(s)
            for v in variants:
                ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, v, iters, C.byref(ms)))
                if rnd: res[(s, v)].append(ms.value * 1e3 / iters)   # round 0 = warm-up
    for s in args.settings:
        for v in variants:
            r = res[(s, v)]
            med = statistics.median(r)
            print("%5d^2 %-34s %-13s median %8.2f us  min %8.2f  max %8.2f   %6.1f GB/s" %
                  (dim, s or "(defaults)", names.get(v, str(v)), med, min(r), max(r), 3.0 * h * w / med / 1e3), flush=True)
    L.tic_dev_free(ctx.handle, d_img); L.tic_dev_free(ctx.handle, d_out)
