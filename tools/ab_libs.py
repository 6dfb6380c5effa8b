"""A/B timing of two BUILDS of the library in one process (boxes differ by +-5 %, builds must be compared on one box): the product library
against another .so of the same C-ABI (e.g. an older kernel built into tools/bin/), the same frame, rounds interleaved.  Warm (one
frame/coefficient pair replayed) and cold (12 rotating pairs).  Usage: python tools/ab_libs.py tools/bin/libother.so [more.so ...] [--dim 4096] [--rounds 7]"""
import argparse, ctypes as C, statistics, sys
sys.path.insert(0, '.')
import numpy as np
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("other", nargs="+")
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--quality", type=int, default=50)
args = ap.parse_args()

def bind(path):
    L = C.CDLL(path)
    for name in ("tic_create", "tic_destroy", "tic_dev_alloc", "tic_dev_free", "tic_memcpy_h2d", "tic_memcpy_d2h", "tic_dctq_dev_timed", "tic_dctq_dev_timed_rotating"):
        res, a = N.SIGNATURES[name]
        fn = getattr(L, name); fn.restype = res; fn.argtypes = a
    return L

h = w = args.dim
import os
libs = {"product": bind(N.LIB_PATH)}
for pth in args.other:
    libs[os.path.basename(pth).replace("lib", "").replace(".so", "")] = bind(pth)
state = {}
pairs = 12
for name, L in libs.items():
    ctx = L.tic_create(0)
    assert ctx
    imgs, outs = [], []
    for k in range(pairs):
        img = np.random.default_rng(1234 + k).integers(0, 256, (h, w), dtype=np.uint8)
        d_img, d_out = C.c_void_p(), C.c_void_p()
        assert L.tic_dev_alloc(ctx, img.size, C.byref(d_img)) == 0 and L.tic_dev_alloc(ctx, img.size * 2, C.byref(d_out)) == 0
        assert L.tic_memcpy_h2d(ctx, d_img, img.ctypes.data, img.size) == 0
        imgs.append(d_img); outs.append(d_out)
    state[name] = (L, ctx, imgs, outs)
ms = C.c_float()
def warm(name, iters):
    L, ctx, imgs, outs = state[name]
    assert L.tic_dctq_dev_timed(ctx, imgs[0], h, w, w, args.quality, outs[0], 2, iters, C.byref(ms)) == 0
    return ms.value * 1e3 / iters
def cold(name, iters):
    L, ctx, imgs, outs = state[name]
    pi = (C.c_void_p * pairs)(*imgs); po = (C.c_void_p * pairs)(*outs)
    assert L.tic_dctq_dev_timed_rotating(ctx, pi, po, pairs, h, w, w, args.quality, 2, iters, C.byref(ms)) == 0
    return ms.value * 1e3 / iters
# same coefficients?
a = np.empty((h // 8) * (w // 8) * 64, np.int16); b = np.empty_like(a)
for name in libs:
    L, ctx, imgs, outs = state[name]
    buf = a if name == "product" else b
    warm(name, 1)
    assert L.tic_memcpy_d2h(ctx, buf.ctypes.data, outs[0], buf.nbytes) == 0
    if name != "product": print("coefficients of %s identical to the product's:" % name, bool(np.array_equal(a, b)))
for name in libs: warm(name, 3000); cold(name, 600)
res = {(n, m): [] for n in libs for m in ("warm", "cold")}
for r in range(args.rounds):
    for name in libs:
        res[(name, "warm")].append(warm(name, args.iters))
        res[(name, "cold")].append(cold(name, args.iters // 4))
for (name, mode), v in res.items():
    print("%-10s %-5s median %.3f us  min %.3f  max %.3f   (%s)" % (name, mode, statistics.median(v), min(v), max(v), " ".join("%.2f" % x for x in v)))
