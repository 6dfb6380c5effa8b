"""Timing-only ablations of the hybrid kernel (outputs of variants >= 10 are wrong by construction)."""
import ctypes as C, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _ablate  # noqa: F401  (experiment build of the library)
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
for dim in (4096, 16384):
    h = w = dim
    img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    ms = C.c_float()
    for name, v in (("full hybrid", 2), ("no arithmetic (LDS+mem)", 10), ("no LDS (VALU+mem)", 11), ("no LDS, no arithmetic", 15), ("no post-pass", 12), ("post-pass: ties only", 13), ("post-pass: exact only", 14), ("post-pass: exact w/o arithmetic", 16), ("compute only (no memory traffic)", 18), ("compute only, no arithmetic (LDS + SALU)", 20), ("one-block-per-lane kernel (variant 40)", 40), ("one-block-per-lane, rare paths off", 41), ("exact kernel", 1)):
        iters = 50 if dim == 4096 else 10
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, v, 5, C.byref(ms)))
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, v, iters, C.byref(ms)))
        us = ms.value * 1e3 / iters
        print("%5d^2 %-42s %9.2f us  %7.1f GB/s (3 B/px)" % (dim, name, us, 3.0 * h * w / us / 1e3))
    L.tic_dev_free(ctx.handle, d_img); L.tic_dev_free(ctx.handle, d_out)
